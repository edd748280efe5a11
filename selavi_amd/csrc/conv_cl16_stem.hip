// 16-bit MFMA path: the STEM convolutions -- Conv3d(3 -> 45, (1,7,7), stride (1,2,2), padding (0,3,3)) of R(2+1)D-18 and
// Conv2d(1 -> 64, 7, 2, 3) of the audio ResNet (torchvision's stems: /root/reference/model.py:93-114 builds them,
// main.py:296-299 runs their backward) -- DIRECTLY from the fp32 N,C,T,H,W clip / spectrogram.
//
// Rounds 1-4 went through a materialised "W-patch" tensor ([N][T][H][Wo][32] bf16 = 1.64 GB at 128 clips x 32 frames:
// slv_to_cl16_wpatch, then a (1,7,1) conv over 32 channels of which 21 are real, and the same tensor again for the weight
// gradient): 0.86 ms to write it, 0.92 ms for the conv that reads it, 1.56 ms for the weight gradient -- against 0.6 GB of
// input and 1.6 GB of output.  Here a workgroup stages the input rows of a BAND of 8 output rows in LDS once, as bf16 pixels
// of 4 channels (8 bytes; channel 3 = 0) behind a 3-pixel left margin, so that the K = 32 slice of one kernel row,
// k = 4 dw + c (dw = 0..7, the 8th column has zero weights), of output pixel ox starts at LDS pixel 2 ox: a 16-byte aligned
// ds_read_b128 per k-group and MFMA B fragment, no im2col tensor anywhere.
//   forward : per 16 positions (2 rows x 8 columns) and kernel row one fragment read feeds MT MFMAs against the weights
//             resident in registers (7 x MT fragments); output through a wave-private LDS stage -> whole 128-byte channel
//             rows to memory; BatchNorm statistics of the rounded outputs as per-lane running sums; persistent workgroups
//             (the next band's pixels are requested before this band's MFMAs), one statistics partial per wave.
//   weight gradient : dW[co][c][dh][dw] = sum over positions of dY[pos][co] * x[c][2 oy + dh - 3][2 ox + dw - 3] as 7 GEMMs
//             (one per kernel row) of M = channels, N = 32 = (dw, c), K = positions: the same staged band, dY by 8 x 8-pixel
//             tiles (double buffered), both operands by ds_read_b64_tr_b16 (K = positions runs ACROSS the rows of either
//             LDS image; the transposing read takes any per-lane address, so the stride-2 windows need no copy); wave w owns
//             kernel rows w and w + 4 (2 x 2 x MT accumulator tiles); every workgroup writes its fp32 partial,
//             cl16_stem_wgrad_reduce_kernel sums them in a fixed order into the reference layout [Cout][Cin][1][7][7].
#include "cl16.hpp"
#include "../../include/selavi_hip.h"

namespace slv {

constexpr int ST_ROWS = 8;                            // output rows of a band
constexpr int ST_IROWS = 2 * ST_ROWS + 5;             // input rows a band reads (stride 2, 7 taps)
constexpr int ST_PITCH = 960;                         // bytes per staged image row: 120 pixels of 8 bytes; 960 = 64 (mod 128),
                                                      // so the two rows of a fragment (2 * pitch = 128 mod 256) use disjoint banks
constexpr int ST_JW = ST_PITCH / 8;                   // staged pixels per row (margin 3 + W + right margin: W <= 112)
constexpr int ST_FIT = (ST_IROWS * ST_JW + 255) / 256;                // 10 pixels per thread and band
constexpr int ST_OROW = 128 + 16;                     // bytes per position of a wave's output stage (64 channels + banks)
constexpr int ST_IMG = ST_IROWS * ST_PITCH;           // 20 160
constexpr int ST_STAGE = 16 * ST_OROW;                // 2 304 per wave
constexpr int ST_LDS_FWD = ST_IMG + 4 * ST_STAGE + 256 * 8;           // + where the pixels past the image go: 31 424

struct StemGeom {
  int N, Cin, T, H, W, Ho, Wo, Cout;
  int nbf, ncg;                                       // bands per frame, 8-column groups per row
  unsigned mnbf, mT, mncg;                            // their reciprocals (2^32 / d + 1: exact quotients by multiply-high; the
};                                                    // band / tile decodes run per band and per 16-position tile)
__device__ __forceinline__ int st_div(int n, unsigned m) { return (int)cl_div((unsigned)n, m); }

// the staged image of a band: pixel idx = tid + 256 it -> (row, j); image column ix = j - 3
struct StemFill {
  int row[ST_FIT];                                    // image row within the band (1 << 20: never inside)
  int off[ST_FIT];                                    // row * W + ix
  int lds[ST_FIT];                                    // byte offset in LDS (pixels past the image: the dump area)
};

__device__ __forceinline__ void stem_fill_init(StemFill& f, int tid, int W) {
#pragma unroll
  for (int it = 0; it < ST_FIT; ++it) {
    const int idx = tid + 256 * it, row = idx / ST_JW, j = idx - row * ST_JW, ix = j - 3;
    const bool in = idx < ST_IROWS * ST_JW;
    f.row[it] = (in && ix >= 0 && ix < W) ? row : (1 << 20);
    f.off[it] = row * W + ix;
    f.lds[it] = in ? row * ST_PITCH + j * 8 : ST_IMG + 4 * ST_STAGE + tid * 8;
  }
}

// requests of band b (frame f, output rows oy0 ..): three fp32 planes per pixel; rows / columns outside the image and
// channels >= Cin come back as zeros (out-of-range offsets)
__device__ __forceinline__ void stem_load(const StemFill& f, const StemGeom& g, __amdgpu_buffer_rsrc_t rx, int b, int total,
                                          float (&r)[ST_FIT][3]) {
  const int fr = st_div(b, g.mnbf), band = b - fr * g.nbf;
  const int n = st_div(fr, g.mT), t = fr - n * g.T;
  const int iy0 = 2 * band * ST_ROWS - 3;
  const int HW = g.H * g.W;
  const int lo = b < total ? -iy0 : (1 << 21), hi = g.H - iy0;       // image rows row with lo <= row < hi exist
  const int p0 = ((n * g.Cin) * g.T + t) * HW + iy0 * g.W;           // (channel 0; channel c: + c T H W)
#pragma unroll
  for (int it = 0; it < ST_FIT; ++it) {
    const bool ok = f.row[it] >= lo && f.row[it] < hi;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const unsigned o = (ok && c < g.Cin) ? (unsigned)(p0 + c * g.T * HW + f.off[it]) * 4u : 0xFFFFFFF0u;
      r[it][c] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx, o, 0, 0));
    }
  }
}

__device__ __forceinline__ void stem_store(const StemFill& f, unsigned char* lds, const float (&r)[ST_FIT][3]) {
#pragma unroll
  for (int it = 0; it < ST_FIT; ++it)
    *(uint2*)(lds + f.lds[it]) = make_uint2(pack_bf2(r[it][0], r[it][1]), pack_bf2(r[it][2], 0.f));
}

// A fragment of kernel row dh and output-channel tile i: row (lane & 15) = channel 16 i + row, k = 8 (lane >> 4) .. + 7 with
// k = 4 dw + c
__device__ __forceinline__ bf16x8 stem_w_fragment(const float* __restrict__ w, const StemGeom& g, int dh, int i, int lane) {
  const int co = i * 16 + (lane & 15), kg = lane >> 4;
  float v[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int dw = 2 * kg + (e >> 2), c = e & 3;
    v[e] = (co < g.Cout && c < g.Cin && dw < 7) ? w[((co * g.Cin + c) * 7 + dh) * 7 + dw] : 0.f;
  }
  const u32x4 p = {pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]), pack_bf2(v[4], v[5]), pack_bf2(v[6], v[7])};
  return __builtin_bit_cast(bf16x8, p);
}

// y [N][T][Ho][Wo][64] bf16; stat_sum / stat_sq [Cout][4 gridDim.x]: one partial per wave
template <int MT>
__global__ __launch_bounds__(256, MT == 3 ? 2 : 1) void cl16_stem_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                              unsigned short* __restrict__ y, float* __restrict__ stat_sum,
                                                              float* __restrict__ stat_sq, StemGeom g, int total) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n = lane & 15, kg = lane >> 4;
  unsigned char* const stage = lds + ST_IMG + wave * ST_STAGE;
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, (int)((unsigned)g.N * g.Cin * g.T * g.H * g.W * 4u), 0x00020000);
  const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc((void*)y, 0, (int)((unsigned)g.N * g.T * g.Ho * g.Wo * 128u), 0x00020000);

  bf16x8 A[7][MT];
#pragma unroll
  for (int dh = 0; dh < 7; ++dh)
#pragma unroll
    for (int i = 0; i < MT; ++i) A[dh][i] = stem_w_fragment(w, g, dh, i, lane);
  // the stage's channel pieces beyond the accumulator tiles (MT = 3: channels 48..63) stay zero
  for (int i = lane * 16; i < ST_STAGE; i += 64 * 16) *(u32x4*)(stage + i) = (u32x4){0u, 0u, 0u, 0u};

  StemFill fill;
  stem_fill_init(fill, tid, g.W);
  float r[ST_FIT][3];
  float stS[MT][4], stQ[MT][4];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int e = 0; e < 4; ++e) stS[i][e] = stQ[i][e] = 0.f;

  // fragment address of this lane: position (pr, pc) = (n >> 3, n & 7) of a 2 x 8 tile, k-group kg -> staged pixel 2 ox + 2 kg
  const int lbase = (2 * (n >> 3)) * ST_PITCH + (2 * (n & 7) + 2 * kg) * 8;

  int b = blockIdx.x;
  stem_load(fill, g, rx, b, total, r);
  for (; b < total; b += gridDim.x) {
    stem_store(fill, lds, r);
    __syncthreads();
    stem_load(fill, g, rx, b + gridDim.x, total, r);  // the next band's pixels: in flight behind this band's MFMAs
    const int fr = st_div(b, g.mnbf), band = b - fr * g.nbf, oy0 = band * ST_ROWS;
    for (int q = wave; q < 4 * g.ncg; q += 4) {
      const int rp = st_div(q, g.mncg), cg = q - rp * g.ncg;
      const unsigned char* src = lds + lbase + rp * (4 * ST_PITCH) + cg * 128;
      f32x4 acc[MT];
#pragma unroll
      for (int i = 0; i < MT; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int dh = 0; dh < 7; ++dh) {
        const bf16x8 bfr = *(const bf16x8*)(src + dh * ST_PITCH);
#pragma unroll
        for (int i = 0; i < MT; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[dh][i], bfr, acc[i], 0, 0, 0);
      }
      // C/D layout: column n = the position, rows 4 kg .. 4 kg + 3 = channels 16 i + 4 kg + e
      const bool live = oy0 + 2 * rp + (n >> 3) < g.Ho && cg * 8 + (n & 7) < g.Wo;
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        unsigned lo = pack_bf2(acc[i][0], acc[i][1]), hi = pack_bf2(acc[i][2], acc[i][3]);
        if (!live) lo = hi = 0u;
        *(uint2*)(stage + n * ST_OROW + i * 32 + kg * 8) = make_uint2(lo, hi);
        const float v[4] = {bf_lo(lo), bf_hi(lo), bf_lo(hi), bf_hi(hi)};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          stS[i][e] += v[e];
          stQ[i][e] = __builtin_fmaf(v[e], v[e], stQ[i][e]);
        }
      }
      // the tile's 16 channel rows of 128 bytes: 128 pieces of 16 bytes, a tile row of 8 pixels = 1 KB contiguous
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int id = h * 64 + lane, pos = id >> 3, pc8 = id & 7;
        const u32x4 v = *(const u32x4*)(stage + pos * ST_OROW + pc8 * 16);
        const int oy = oy0 + 2 * rp + (pos >> 3), ox = cg * 8 + (pos & 7);
        const bool ok = oy < g.Ho && ox < g.Wo;
        const unsigned off = ((unsigned)((fr * g.Ho + oy) * g.Wo + ox)) * 128u + pc8 * 16u;
        __builtin_amdgcn_raw_buffer_store_b128(v, ry, ok ? off : 0xFFFFFFF0u, 0, CL_NT);
      }
    }
    __syncthreads();                                  // every wave has read the image: the next band may overwrite it
  }
  if (stat_sum) {
    const int nblk = 4 * gridDim.x, slot = 4 * blockIdx.x + wave;
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float a = row16_sum(stS[i][e]), qq = row16_sum(stQ[i][e]);
        const int c = i * 16 + 4 * kg + e;
        if (n == 0 && c < g.Cout) {
          stat_sum[(size_t)c * nblk + slot] = a;
          stat_sq[(size_t)c * nblk + slot] = qq;
        }
      }
  }
}

constexpr int ST_GP = 128 + 32;                       // bytes per position of a staged dY tile (= 32 mod 64: transposing reads)
constexpr int ST_GT = 64 * ST_GP;                     // 10 240 per tile
constexpr int ST_LDS_WG = ST_IMG + 2 * ST_GT + 256 * 8;               // 42 688

// part [gridDim.x][16 MT][7][32] fp32: this workgroup's sum over its bands; column k = 4 dw + c of kernel row dh
// APPLY: dy is the gradient w.r.t. the ACTIVATED output of the conv's BatchNorm; the staged tile is what slv_cl16_bn_bwd_apply
// would make of it (A1 * mask * g + A2 + A3 * y with y = the conv's raw output, b5 = {s, h, A1, A2, A3}[Cout]), bit for bit --
// the stem's first conv has no backward-data launch, so its weight gradient was that pass's only reader
template <int MT, bool APPLY>
__global__ __launch_bounds__(256, MT == 3 ? 2 : 1) void cl16_stem_wgrad_kernel(const float* __restrict__ x,
                                                                              const unsigned short* __restrict__ dy,
                                                                              float* __restrict__ part, StemGeom g, int total,
                                                                              const unsigned short* __restrict__ yraw,
                                                                              const float* __restrict__ b5, int relu) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  typedef __attribute__((address_space(3))) s16x4* lds_s16x4;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 15, fk = lane >> 4;
  unsigned char* const gl = lds + ST_IMG;             // [2][ST_GT]
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, (int)((unsigned)g.N * g.Cin * g.T * g.H * g.W * 4u), 0x00020000);
  const __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc((void*)dy, 0, (int)((unsigned)g.N * g.T * g.Ho * g.Wo * 128u), 0x00020000);
  const __amdgpu_buffer_rsrc_t ryr = __builtin_amdgcn_make_buffer_rsrc((void*)(APPLY ? yraw : dy), 0, (int)((unsigned)g.N * g.T * g.Ho * g.Wo * 128u), 0x00020000);
  float cs[8], ch[8], c1[8], c2[8], c3[8];            // this thread's 8 channels (piece tid & 7 of every tile position it stages)
  if constexpr (APPLY) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int c = (tid & 7) * 8 + e;
      const bool ok = c < g.Cout;
      cs[e] = ok ? b5[c] : 0.f;
      ch[e] = ok ? b5[g.Cout + c] : 0.f;
      c1[e] = ok ? b5[2 * g.Cout + c] : 0.f;
      c2[e] = ok ? b5[3 * g.Cout + c] : 0.f;
      c3[e] = ok ? b5[4 * g.Cout + c] : 0.f;
    }
  }
  StemFill fill;
  stem_fill_init(fill, tid, g.W);
  float r[ST_FIT][3];
  // dY tile: piece id = tid + 256 h of 512: position id >> 3 = (row, column) of the 8 x 8 tile, 16-byte piece id & 7
  u32x4 gr[2], yr[2];
  unsigned gok = 0;                                   // bit h: piece h of the staged tile is a pixel of the image
  auto load_g = [&](int b, int cg) __attribute__((always_inline)) {
    const int f = st_div(b, g.mnbf), band = b - f * g.nbf, oy0 = band * ST_ROWS;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int id = tid + 256 * h, pos = id >> 3, oy = oy0 + (pos >> 3), ox = cg * 8 + (pos & 7);
      const bool ok = b < total && cg < g.ncg && oy < g.Ho && ox < g.Wo;
      const unsigned off = ((unsigned)((f * g.Ho + oy) * g.Wo + ox)) * 128u + (id & 7) * 16u;
      gr[h] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rg, ok ? off : 0xFFFFFFF0u, 0, 0));
      if constexpr (APPLY) {
        yr[h] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(ryr, ok ? off : 0xFFFFFFF0u, 0, 0));
        gok = (gok & ~(1u << h)) | ((unsigned)ok << h);
      }
    }
  };
  auto store_g = [&](int par) __attribute__((always_inline)) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int id = tid + 256 * h;
      u32x4 v = gr[h];
      if constexpr (APPLY) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          float o2[2];
#pragma unroll
          for (int e = 0; e < 2; ++e) {
            const int k = 2 * i + e;
            const float xx = e ? bf_hi(yr[h][i]) : bf_lo(yr[h][i]);
            float gg = e ? bf_hi(gr[h][i]) : bf_lo(gr[h][i]);
            if (relu && !(bn_affine(xx, cs[k], ch[k]) > 0.f)) gg = 0.f;
            o2[e] = bn_bwd_apply1(gg, xx, c1[k], c2[k], c3[k]);
          }
          v[i] = ((gok >> h) & 1u) ? pack_bf2(o2[0], o2[1]) : 0u;     // (outside the image: no gradient, not A2)
        }
      }
      *(u32x4*)(gl + par * ST_GT + (id >> 3) * ST_GP + (id & 7) * 16) = v;
    }
  };
  // the transposing reads: lane (fr, fk) supplies the address of position 4 fk + (fr >> 2) (+ 16) and 8 bytes (4 columns) at
  // 4 (fr & 3); it receives column fr at the positions 4 fk .. 4 fk + 3 (+ 16).  Position p of a K step ks = tile pixel
  // (4 ks + (p >> 3), p & 7)
  const int p0 = 4 * fk + (fr >> 2);
  const int glane = p0 * ST_GP + 8 * (fr & 3);                                   // dY: columns = channels
  const int xlane = (2 * (p0 >> 3)) * ST_PITCH + (2 * (p0 & 7) + (fr & 3)) * 8;  // image: columns = (dw, c), pixel 2 ox + dw
  auto rd2 = [&](const unsigned char* lo_p, int hi_off) __attribute__((always_inline)) {
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(lo_p));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(lo_p + hi_off));
    return tr_pair(lo, hi);
  };
  const int ndh = wave < 3 ? 2 : 1;                   // kernel rows wave, wave + 4
  f32x4 acc[2][2][MT];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int i = 0; i < MT; ++i) acc[a][nt][i] = (f32x4){0.f, 0.f, 0.f, 0.f};

  int par = 0;
  int b = blockIdx.x;
  stem_load(fill, g, rx, b, total, r);
  for (; b < total; b += gridDim.x) {
    stem_store(fill, lds, r);
    load_g(b, 0);
    stem_load(fill, g, rx, b + gridDim.x, total, r);  // the next band's pixels: in flight behind this band's tiles
    for (int cg = 0; cg < g.ncg; ++cg) {
      store_g(par);
      __syncthreads();
      load_g(b, cg + 1);                              // (past the band's last tile: nothing is requested)
      const unsigned char* gt = gl + par * ST_GT + glane;
      const unsigned char* xt = lds + xlane + cg * 128;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        bf16x8 af[MT];
#pragma unroll
        for (int i = 0; i < MT; ++i) af[i] = rd2(gt + ks * 32 * ST_GP + i * 32, 16 * ST_GP);
#pragma unroll
        for (int a = 0; a < 2; ++a) {
          if (a < ndh) {
            const int dh = wave + 4 * a;
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
              // K step ks = tile rows 4 ks .. 4 ks + 3 (image rows 8 ks + dh ..); the upper half of the K step two tile rows on
              const bf16x8 bfr = rd2(xt + (8 * ks + dh) * ST_PITCH + nt * 32, 4 * ST_PITCH);
#pragma unroll
              for (int i = 0; i < MT; ++i) acc[a][nt][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr, acc[a][nt][i], 0, 0, 0);
            }
          }
        }
      }
      par ^= 1;
    }
    __syncthreads();                                  // every wave has read the image: the next band may overwrite it
  }
  // C/D layout: column fr = (dw, c) within the column tile, rows 4 fk + e = channels
  float* pw = part + (size_t)blockIdx.x * (16 * MT * 7 * 32);
#pragma unroll
  for (int a = 0; a < 2; ++a)
    if (a < ndh) {
      const int dh = wave + 4 * a;
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
          for (int e = 0; e < 4; ++e) pw[((i * 16 + 4 * fk + e) * 7 + dh) * 32 + nt * 16 + fr] = acc[a][nt][i][e];
    }
}

// dw [Cout][Cin][7][7] = sum over the workgroups' partials, in index order
__global__ __launch_bounds__(256) void cl16_stem_wgrad_reduce_kernel(const float* __restrict__ part, float* __restrict__ dw,
                                                                    int nwg, int mrows, int Cout, int Cin) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= Cout * Cin * 49) return;
  const int kw = idx % 7, kh = (idx / 7) % 7, c = (idx / 49) % Cin, co = idx / (49 * Cin);
  const float* p = part + ((size_t)co * 7 + kh) * 32 + 4 * kw + c;
  float s = 0.f;
  for (int j = 0; j < nwg; ++j) s += p[(size_t)j * mrows * 7 * 32];
  dw[idx] = s;
}

static int stem_grid(long long total) {
  static const int blocks = []() {
    const char* e = getenv("SELAVI_CL16_STEM_BLOCKS");
    return e ? atoi(e) : 512;                         // two persistent workgroups per CU
  }();
  return (int)(total < blocks ? total : blocks);
}

static bool stem_enabled() {
  static const bool on = []() {
    const char* e = getenv("SELAVI_CL16_STEM");
    return !(e && e[0] == '0');
  }();
  return on;
}

static bool stem_geom(StemGeom& g, int N, int Cin, int T, int H, int W, int Cout) {
  if (N < 1 || Cin < 1 || Cin > 3 || T < 1 || H < 1 || W < 1 || W > ST_JW - 8 || Cout <= 32 || Cout > 64) return false;   // (64 stored channels)
  g.N = N, g.Cin = Cin, g.T = T, g.H = H, g.W = W, g.Cout = Cout;
  g.Ho = (H + 6 - 7) / 2 + 1, g.Wo = (W + 6 - 7) / 2 + 1;
  g.nbf = (g.Ho + ST_ROWS - 1) / ST_ROWS, g.ncg = (g.Wo + 7) / 8;
  g.mnbf = cl_recip((unsigned)g.nbf), g.mT = cl_recip((unsigned)T), g.mncg = cl_recip((unsigned)g.ncg);
  if (2 * (g.ncg * 8 - 1) + 8 > ST_JW) return false;                   // the last column group's fragments stay inside a staged row
  if ((long long)N * Cin * T * H * W * 4 >= 0xFFFFFFF0LL || (long long)N * T * g.Ho * g.Wo * 128 >= 0xFFFFFFF0LL) return false;
  if (((long long)N * T * g.nbf + 4096) * (g.nbf > T ? g.nbf : T) >= 0xFFFFFFFFLL) return false;      // (exact multiply-high decodes)
  return true;
}

}  // namespace slv

extern "C" {

int32_t slv_cl16_stem_ok(int N, int Cin, int T, int H, int W, int Cout, int kh, int kw, int sh, int sw, int ph, int pw) {
  slv::StemGeom g;
  return slv::stem_enabled() && kh == 7 && kw == 7 && sh == 2 && sw == 2 && ph == 3 && pw == 3 &&
                 slv::stem_geom(g, N, Cin, T, H, W, Cout)
             ? 1
             : 0;
}

int32_t slv_cl16_stem_nblk(int N, int Cin, int T, int H, int W, int Cout) {
  slv::StemGeom g;
  if (!slv::stem_geom(g, N, Cin, T, H, W, Cout)) return 0;
  return 4 * slv::stem_grid((long long)N * T * g.nbf);
}

int slv_cl16_stem_fwd(const float* x, const float* w, void* y_bf16, float* stat_sum, float* stat_sq, int N, int Cin, int T,
                      int H, int W, int Cout, slv_stream_t stream) {
  using namespace slv;
  StemGeom g;
  SLV_CHECK_ARG(x && w && y_bf16 && (!stat_sum == !stat_sq) && stem_geom(g, N, Cin, T, H, W, Cout),
                "7 x 7 stride-2 stem over <= 3 channels, W <= 112, 33..64 output channels");
  const int total = N * T * g.nbf, grid = stem_grid(total);
  if (Cout <= 48)
    hipLaunchKernelGGL((cl16_stem_fwd_kernel<3>), dim3(grid), dim3(256), ST_LDS_FWD, (hipStream_t)stream, x, w,
                       (unsigned short*)y_bf16, stat_sum, stat_sq, g, total);
  else
    hipLaunchKernelGGL((cl16_stem_fwd_kernel<4>), dim3(grid), dim3(256), ST_LDS_FWD, (hipStream_t)stream, x, w,
                       (unsigned short*)y_bf16, stat_sum, stat_sq, g, total);
  SLV_LAUNCH_CHECK();
  return 0;
}

size_t slv_cl16_stem_wgrad_ws_bytes(int N, int Cin, int T, int H, int W, int Cout) {
  slv::StemGeom g;
  if (!slv::stem_geom(g, N, Cin, T, H, W, Cout)) return 0;
  return (size_t)slv::stem_grid((long long)N * T * g.nbf) * (Cout <= 48 ? 48 : 64) * 7 * 32 * sizeof(float);
}

int slv_cl16_stem_wgrad(const float* x, const void* dy_bf16, float* dw, float* ws, size_t ws_bytes, int N, int Cin, int T, int H,
                        int W, int Cout, const void* y_bf16, const float* bwd5, int relu, slv_stream_t stream) {
  using namespace slv;
  StemGeom g;
  SLV_CHECK_ARG(x && dy_bf16 && dw && ws && stem_geom(g, N, Cin, T, H, W, Cout),
                "7 x 7 stride-2 stem over <= 3 channels, W <= 112, 33..64 output channels");
  SLV_CHECK_ARG(ws_bytes >= slv_cl16_stem_wgrad_ws_bytes(N, Cin, T, H, W, Cout), "workspace too small");
  SLV_CHECK_ARG(!y_bf16 == !bwd5, "y_bf16 and bwd5 come together");
  const int total = N * T * g.nbf, grid = stem_grid(total), mrows = Cout <= 48 ? 48 : 64;
#define SLV_STEM_WG(MT_, AP_)                                                                                                  \
  hipLaunchKernelGGL((cl16_stem_wgrad_kernel<MT_, AP_>), dim3(grid), dim3(256), ST_LDS_WG, (hipStream_t)stream, x,             \
                     (const unsigned short*)dy_bf16, ws, g, total, (const unsigned short*)y_bf16, bwd5, relu)
  if (Cout <= 48) {
    if (y_bf16) SLV_STEM_WG(3, true);
    else SLV_STEM_WG(3, false);
  } else {
    if (y_bf16) SLV_STEM_WG(4, true);
    else SLV_STEM_WG(4, false);
  }
#undef SLV_STEM_WG
  SLV_LAUNCH_CHECK();
  hipLaunchKernelGGL(cl16_stem_wgrad_reduce_kernel, dim3((Cout * Cin * 49 + 255) / 256), dim3(256), 0, (hipStream_t)stream, ws,
                     dw, grid, mrows, Cout, Cin);
  SLV_LAUNCH_CHECK();
  return 0;
}

}  // extern "C"
