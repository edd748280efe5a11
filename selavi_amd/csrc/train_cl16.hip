// Training kernels of the 16-bit MFMA path (BASELINE configs[4]; the reference trains its convs in half precision
// through apex O1, main.py:151-153,296-299) around csrc/conv_cl16.hip, which runs the forward conv (train mode:
// BatchNorm + ReLU of the producer applied on load, batch statistics from the epilogue) and the backward-data conv
// (same kernel on transposed weights, one launch per stride-parity class):
//   * weight re-layouts from the fp32 master weights (per step): forward + backward-data bf16 layouts
//   * the WEIGHT GRADIENT: dW[co][(tap, ci)] = sum_pos dY[pos][co] * act(X)[pos*stride + tap - pad][ci].  Both operands
//     are channels-last, i.e. the contraction index (position) is the SLOW dimension of both: the tiles are staged in LDS
//     position-major as they are loaded and the MFMA fragments (8 consecutive k per lane) come out of
//     ds_read_b64_tr_b16, the LDS transpose read of gfx950.  fp32 partial tiles per K-slice, fixed-order reduce into the
//     reference's fp32 [Cout][Cin][kt][kh][kw] layout (the optimizer keeps fp32 master weights).
//   * channels-last bf16 versions of the HBM-bound BatchNorm kernels of csrc/elementwise.hip (block tail, backward
//     reductions, backward apply) and the average-pool backward.
#include "cl16.hpp"
#include "../../include/selavi_hip.h"

namespace slv {

// ------------------------------------------------------------------------------------------ weight layouts
// w fp32 [Cout][Cin][taps] ->
//   wf bf16 [taps][Cin_p/32][MrowsF][32]   wf(tap, kc, m, j) = w[m][kc*32 + j][tap]          (forward: rows = cout)
//   wt bf16 [taps][Cout_p/32][MrowsD][32]  wt(tap, kc, m, j) = w[kc*32 + j][m][tap]          (backward data: rows = cin)
// patch_kw > 0 (stem, csrc/conv_cl16.hip:to_cl16_wpatch_kernel): the forward conv is a (1, kh, 1) conv over the
// 32-channel W-patch layout, channel j = dw*Cin + c:  wf(a, 0, m, j) = w[m][c][0][a][dw].
__global__ __launch_bounds__(256) void cl16_w_transform_kernel(const float* __restrict__ w, unsigned short* __restrict__ wf,
                                                               unsigned short* __restrict__ wt, int Cout, int Cin,
                                                               int taps, int Cin_p, int Cout_p, int MrowsF, int MrowsD,
                                                               int patch_kw, unsigned nf, unsigned nt) {
  const unsigned idx = blockIdx.x * 256 + threadIdx.x;
  if (idx < nf) {
    const unsigned j = idx & 31, m = (idx >> 5) % MrowsF, r = (idx >> 5) / MrowsF;
    float v = 0.f;
    if (patch_kw > 0) {
      const unsigned a = r;                       // kcs == 1
      const unsigned dw = j / Cin, c = j - dw * Cin;
      if (m < (unsigned)Cout && dw < (unsigned)patch_kw) v = w[((size_t)(m * Cin + c) * taps + a) * patch_kw + dw];
    } else {
      const unsigned kcs = Cin_p >> 5, kc = r % kcs, tap = r / kcs, c = kc * 32 + j;
      if (m < (unsigned)Cout && c < (unsigned)Cin) v = w[((size_t)m * Cin + c) * taps + tap];
    }
    wf[idx] = f2bf(v);
  } else if (idx - nf < nt) {
    const unsigned i2 = idx - nf;
    const unsigned j = i2 & 31, m = (i2 >> 5) % MrowsD, r = (i2 >> 5) / MrowsD;
    const unsigned kcs = Cout_p >> 5, kc = r % kcs, tap = r / kcs, co = kc * 32 + j;
    float v = 0.f;
    if (co < (unsigned)Cout && m < (unsigned)Cin) v = w[((size_t)co * Cin + m) * taps + tap];
    wt[i2] = f2bf(v);
  }
}

// The same for MANY layers in one launch (slv_cl16_w_transform_jobs): a job = the arguments of one layer's launch, the table in
// device memory (built once per model and input shape by selavi_amd/ops16.py: parameter and layout buffers are persistent),
// blockIdx.y = job, grid-stride over the job's elements.
struct ClWJob {
  const float* w;
  unsigned short* wf;
  unsigned short* wt;
  int Cout, Cin, taps, Cin_p, Cout_p, MrowsF, MrowsD, patch_kw;
  unsigned nf, nt;
  unsigned first, count;      // the elements (of the nf + nt) this job makes: equal jobs, whatever the layer's size
};
static_assert(sizeof(ClWJob) == 72, "18 int32 words: mirrored by selavi_amd/ops16.py");
__global__ __launch_bounds__(256) void cl16_w_transform_jobs_kernel(const ClWJob* __restrict__ jobs) {
  const ClWJob j = jobs[blockIdx.y];
  const unsigned total = j.first + j.count;
  for (unsigned idx = j.first + blockIdx.x * 256 + threadIdx.x; idx < total; idx += gridDim.x * 256) {
    if (idx < j.nf) {
      const unsigned jj = idx & 31, m = (idx >> 5) % j.MrowsF, r = (idx >> 5) / j.MrowsF;
      float v = 0.f;
      if (j.patch_kw > 0) {
        const unsigned a = r;
        const unsigned dw = jj / j.Cin, c = jj - dw * j.Cin;
        if (m < (unsigned)j.Cout && dw < (unsigned)j.patch_kw) v = j.w[((size_t)(m * j.Cin + c) * j.taps + a) * j.patch_kw + dw];
      } else {
        const unsigned kcs = j.Cin_p >> 5, kc = r % kcs, tap = r / kcs, c = kc * 32 + jj;
        if (m < (unsigned)j.Cout && c < (unsigned)j.Cin) v = j.w[((size_t)m * j.Cin + c) * j.taps + tap];
      }
      j.wf[idx] = f2bf(v);
    } else {
      const unsigned i2 = idx - j.nf;
      const unsigned jj = i2 & 31, m = (i2 >> 5) % j.MrowsD, r = (i2 >> 5) / j.MrowsD;
      const unsigned kcs = j.Cout_p >> 5, kc = r % kcs, tap = r / kcs, co = kc * 32 + jj;
      float v = 0.f;
      if (co < (unsigned)j.Cout && m < (unsigned)j.Cin) v = j.w[((size_t)co * j.Cin + m) * j.taps + tap];
      j.wt[i2] = f2bf(v);
    }
  }
}

// ------------------------------------------------------------------------------------------ weight gradient
struct ClWgrad {                      // int32 x CLW_WORDS, mirrored by selavi_amd/ops16.py
  int N;
  int Ti, Hi, Wi, Cin_p, Cin;         // X [N][Ti][Hi][Wi][Cin_p]
  int To, Ho, Wo, Cout_p;             // dY [N][To][Ho][Wo][Cout_p]
  int st, sh, sw, pt, ph, pw;
  int kt, kh, kw;
  int Ncols;                          // kt*kh*kw * Cin_p: GEMM column = tap * Cin_p + ci
  int mtiles, ntiles;                 // of the (BM, BN) the launch is instantiated for
  int kslices, kper;                  // positions per K-slice (multiple of 32)
};
constexpr int CLW_WORDS = sizeof(ClWgrad) / 4;

// K mapping of one 32-position step: lane group g = lane >> 4 contracts the tile rows {4g..4g+3} and {16+4g..16+4g+3}
// (any bijection works as long as both operands use it; this one makes the 8 rows a half wave touches per transpose
// read consecutive, and with a row stride == 32 (mod 64) bytes they fall into 8 different 32-byte bank groups).
template <int WM, int WN, int PRO>
__global__ __launch_bounds__(256, 2) void cl16_wgrad_kernel(const unsigned short* __restrict__ dy,
                                                            const unsigned short* __restrict__ x,
                                                            const float* __restrict__ in_ss, float* __restrict__ part,
                                                            ClWgrad g, FastDiv dWo, FastDiv dHo, FastDiv dTo) {
  constexpr int BM = 32 * WM, BN = 32 * WN;
  constexpr int SA = BM * 2 + 32, SB = BN * 2 + 32;             // row strides in bytes, (S / 32) odd
  constexpr int APC = BM / 8, BPC = BN / 8;                     // 16-byte pieces per row
  constexpr int AIT = (32 * APC + 255) / 256, BIT = (32 * BPC + 255) / 256;
  constexpr int STAGE = 32 * (SA + SB);
  __shared__ __attribute__((aligned(16))) unsigned char lds_raw[2 * STAGE + (PRO ? 2 * 1152 * 4 : 0)];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // XCD-aware bijective remap: consecutive units (the tiles of one K-slice read the same rows) share an XCD's L2
  const unsigned total = gridDim.x, q8 = total >> 3, r8 = total & 7, xcd = blockIdx.x & 7, loc = blockIdx.x >> 3;
  const unsigned unit = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + loc;
  const unsigned tiles = (unsigned)(g.mtiles * g.ntiles);
  const unsigned slice = unit / tiles, tile = unit - slice * tiles;
  const int m0 = (int)(tile / g.ntiles) * BM, n0 = (int)(tile % g.ntiles) * BN;
  const unsigned P = (unsigned)g.N * g.To * g.Ho * g.Wo;
  const unsigned k_lo = slice * (unsigned)g.kper, k_hi = min(k_lo + (unsigned)g.kper, P);
  const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc((void*)dy, 0, (int)(P * (unsigned)g.Cout_p * 2u), 0x00020000);
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(
      (void*)x, 0, (int)((unsigned)g.N * g.Ti * g.Hi * g.Wi * g.Cin_p * 2u), 0x00020000);
  float* pro = (float*)(lds_raw + 2 * STAGE);
  if constexpr (PRO == 1) {
    for (int i = tid; i < 2 * g.Cin_p; i += 256) {
      const int c = i % g.Cin_p, which = i / g.Cin_p;
      pro[i] = c < g.Cin ? in_ss[which * g.Cin + c] : 0.f;
    }
  }
  // ---- per-thread constants of the loaders
  int arow[AIT];
  unsigned aoff[AIT];                      // byte offset inside a dY row, or ~0 when the piece is outside the tensor
#pragma unroll
  for (int i = 0; i < AIT; ++i) {
    const int pc = tid + 256 * i;
    arow[i] = pc / APC;
    const int c = m0 + (pc % APC) * 8;
    aoff[i] = (pc < 32 * APC && c < g.Cout_p) ? (unsigned)c * 2u : 0xFFFFFFFFu;
  }
  int brow[BIT], bdt[BIT], bdh[BIT], bdw[BIT], bci[BIT];
  bool bval[BIT];
#pragma unroll
  for (int i = 0; i < BIT; ++i) {
    const int pc = tid + 256 * i;
    brow[i] = pc / BPC;
    const int col = n0 + (pc % BPC) * 8;
    bval[i] = pc < 32 * BPC && col < g.Ncols;
    const int tap = col / g.Cin_p;
    bci[i] = col - tap * g.Cin_p;
    bdw[i] = tap % g.kw - g.pw;
    bdh[i] = (tap / g.kw) % g.kh - g.ph;
    bdt[i] = tap / (g.kw * g.kh) - g.pt;
  }
  u32x4 ra[AIT], rb[BIT];
  bool okb[BIT];
  auto gload = [&](unsigned k0) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < AIT; ++i) {
      const unsigned p = k0 + arow[i];
      const unsigned off = (p < k_hi && aoff[i] != 0xFFFFFFFFu) ? p * (unsigned)(g.Cout_p * 2) + aoff[i] : 0xFFFFFFF0u;
      ra[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(ry, off, 0, 0));
    }
#pragma unroll
    for (int i = 0; i < BIT; ++i) {
      const unsigned p = k0 + brow[i];
      unsigned q = fdiv(p, dWo);
      const int wo = p - q * g.Wo;
      unsigned q2 = fdiv(q, dHo);
      const int ho = q - q2 * g.Ho;
      const unsigned n = fdiv(q2, dTo);
      const int to = q2 - n * g.To;
      const int t = to * g.st + bdt[i], h = ho * g.sh + bdh[i], w_ = wo * g.sw + bdw[i];
      const bool ok = bval[i] && p < k_hi && (unsigned)t < (unsigned)g.Ti && (unsigned)h < (unsigned)g.Hi &&
                      (unsigned)w_ < (unsigned)g.Wi;
      okb[i] = ok;
      const unsigned off = (((n * g.Ti + t) * g.Hi + h) * g.Wi + w_) * (unsigned)(g.Cin_p * 2) + (unsigned)bci[i] * 2u;
      rb[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, ok ? off : 0xFFFFFFF0u, 0, 0));
    }
  };
  auto lstore = [&](int buf) __attribute__((always_inline)) {
    unsigned char* A = lds_raw + buf * STAGE;
    unsigned char* B = A + 32 * SA;
#pragma unroll
    for (int i = 0; i < AIT; ++i) {
      const int pc = tid + 256 * i;
      if (pc < 32 * APC) *(u32x4*)(A + arow[i] * SA + (pc % APC) * 16) = ra[i];
    }
#pragma unroll
    for (int i = 0; i < BIT; ++i) {
      const int pc = tid + 256 * i;
      if (pc < 32 * BPC) {
        u32x4 v = rb[i];
        if constexpr (PRO == 1) {
          float s[8], h[8];
          const float* sp = pro + bci[i];
          *(f32x4*)s = *(const f32x4*)sp;
          *(f32x4*)(s + 4) = *(const f32x4*)(sp + 4);
          *(f32x4*)h = *(const f32x4*)(sp + g.Cin_p);
          *(f32x4*)(h + 4) = *(const f32x4*)(sp + g.Cin_p + 4);
          const u32x4 t = affine_relu8(v, s, h);
          v = okb[i] ? t : (u32x4){0u, 0u, 0u, 0u};
        }
        *(u32x4*)(B + brow[i] * SB + (pc % BPC) * 16) = v;
      }
    }
  };
  f32x4 acc[WM][WN];
#pragma unroll
  for (int i = 0; i < WM; ++i)
#pragma unroll
    for (int j = 0; j < WN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const int wm = wave >> 1, wn = wave & 1;
  const int fg = lane >> 4, fi = lane & 15;
  // transpose-read address of this lane: row 4*fg + (fi >> 2) (+16 for the second half), 4 columns at 4*(fi & 3)
  const int fa = (4 * fg + (fi >> 2)) * SA + (wm * WM * 16 + 4 * (fi & 3)) * 2;
  const int fb = (4 * fg + (fi >> 2)) * SB + (wn * WN * 16 + 4 * (fi & 3)) * 2;
  typedef __attribute__((address_space(3))) s16x4* lds_s16x4;
  if (k_lo < k_hi) {
    gload(k_lo);
    if constexpr (PRO == 1) __syncthreads();
    lstore(0);
    __syncthreads();
  }
  int buf = 0;
  for (unsigned k0 = k_lo; k0 < k_hi; k0 += 32) {
    const bool more = k0 + 32 < k_hi;
    if (more) gload(k0 + 32);
    const unsigned char* A = lds_raw + buf * STAGE;
    const unsigned char* B = A + 32 * SA;
    bf16x8 bfr[WN];
#pragma unroll
    for (int j = 0; j < WN; ++j) {
      const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(B + fb + j * 32));
      const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(B + fb + j * 32 + 16 * SB));
      bfr[j] = tr_pair(lo, hi);
    }
    bf16x8 afr[WM];                                    // all fragment reads of the K-step before its first MFMA
#pragma unroll
    for (int i = 0; i < WM; ++i) {
      const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(A + fa + i * 32));
      const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(A + fa + i * 32 + 16 * SA));
      afr[i] = tr_pair(lo, hi);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
      for (int j = 0; j < WN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afr[i], bfr[j], acc[i][j], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    if (more) lstore(buf ^ 1);
    __syncthreads();
    buf ^= 1;
  }
  // ---- partial tile of this K-slice: part[slice][mtiles*BM][ntiles*BN] fp32.  C/D: col = lane & 15 (n), rows
  // (lane >> 4) * 4 + r (m): 64-byte runs per lane group.
  const size_t ldp = (size_t)g.ntiles * BN;
  float* pt = part + ((size_t)slice * g.mtiles * BM + m0 + wm * WM * 16) * ldp + n0 + wn * WN * 16;
#pragma unroll
  for (int i = 0; i < WM; ++i)
#pragma unroll
    for (int j = 0; j < WN; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) pt[(size_t)(i * 16 + fg * 4 + r) * ldp + j * 16 + fi] = acc[i][j][r];
}

// dw[co][ci][tap] = sum_slices part[s][co][tap*Cin_p + ci]  (s ascending: fixed order); patch_kw > 0: the stem's patch
// columns (a*32 + dw*Cin + c) go back to w[co][c][0][a][dw].  One thread per partial-tile COLUMN (consecutive threads
// read consecutive floats of every slice: coalesced; the first version walked the output order and read with a stride
// of Cin_p floats -- 74 us per call, 2.7 ms of a 16-clip step); the scattered 4-byte writes are the small side.
__global__ __launch_bounds__(256) void cl16_wgrad_reduce_kernel(const float* __restrict__ part, float* __restrict__ dw,
                                                                int Cout, int Cin, int taps, int Cin_p, int slices,
                                                                size_t slice_stride, size_t ldp, int patch_kw,
                                                                unsigned ncols /* taps * Cin_p */) {
  const unsigned col = blockIdx.x * 256 + threadIdx.x, co = blockIdx.y;
  if (col >= ncols) return;
  const unsigned tap = col / Cin_p, cc = col - tap * Cin_p;
  size_t out;
  if (patch_kw > 0) {                     // cc = dw*Cin + c
    const unsigned dwi = cc / Cin, c = cc - dwi * Cin;
    if (dwi >= (unsigned)patch_kw) return;
    out = (((size_t)co * Cin + c) * taps + tap) * patch_kw + dwi;
  } else {
    if (cc >= (unsigned)Cin) return;
    out = ((size_t)co * Cin + cc) * taps + tap;
  }
  const float* p = part + (size_t)co * ldp + col;
  float s4[4] = {0.f, 0.f, 0.f, 0.f};                 // four interleaved chains (the loads of one chain are dependent adds),
  int i = 0;                                          // combined in a fixed order
  for (; i + 4 <= slices; i += 4) {
#pragma unroll
    for (int u = 0; u < 4; ++u) s4[u] += p[(size_t)(i + u) * slice_stride];
  }
  for (; i < slices; ++i) s4[i & 3] += p[(size_t)i * slice_stride];
  dw[out] = (s4[0] + s4[1]) + (s4[2] + s4[3]);
}

// ------------------------------------------------------------------------------------------ audio stem pooling
// relu(bn(x)) -> MaxPool2d(3, stride 2, padding 1) on [N][H][W][Cp] bf16 (the audio trunk's stem: torchvision ResNet
// conv1-bn1-relu-maxpool, /root/reference/model.py:114-132), one thread per (output pixel, 8 channels).  The pooled value
// is the bf16-rounded activation (what a conv prologue would have fed the MFMAs); idx = the winning tap 0..8, first
// maximum wins like torch.
__global__ __launch_bounds__(256) void cl16_bnrelu_maxpool_fwd_kernel(const unsigned short* __restrict__ x,
                                                                     const float* __restrict__ ss,
                                                                     unsigned short* __restrict__ out,
                                                                     unsigned char* __restrict__ idx, int C, int Cp, int H,
                                                                     int W, int Ho, int Wo, unsigned total) {
  const unsigned i = blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const unsigned pc = Cp >> 3, piece = i % pc, o = i / pc;
  const int wo = (int)(o % (unsigned)Wo), ho = (int)((o / (unsigned)Wo) % (unsigned)Ho);
  const unsigned n = o / (unsigned)(Wo * Ho);
  const int c0 = piece * 8;
  float s[8], h[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const bool ok = c0 + j < C;
    s[j] = ok ? ss[c0 + j] : 0.f;
    h[j] = ok ? ss[C + c0 + j] : 0.f;
  }
  unsigned best[4] = {0u, 0u, 0u, 0u};
  unsigned char bi[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  bool any = false;
  for (int dh = 0; dh < 3; ++dh) {
    const int hh = ho * 2 - 1 + dh;
    if (hh < 0 || hh >= H) continue;
    for (int dw = 0; dw < 3; ++dw) {
      const int ww = wo * 2 - 1 + dw;
      if (ww < 0 || ww >= W) continue;
      const u32x4 v = affine_relu8(*(const u32x4*)(x + ((size_t)(n * H + hh) * W + ww) * Cp + c0), s, h);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const unsigned cur = (v[j >> 1] >> (16 * (j & 1))) & 0xFFFFu, old = (best[j >> 1] >> (16 * (j & 1))) & 0xFFFFu;
        if (!any || cur > old) {                 // activated values are >= +0: their bf16 patterns order like unsigned ints
          best[j >> 1] = (best[j >> 1] & ~(0xFFFFu << (16 * (j & 1)))) | (cur << (16 * (j & 1)));
          bi[j] = (unsigned char)(dh * 3 + dw);
        }
      }
      any = true;
    }
  }
  *(u32x4*)(out + (size_t)o * Cp + c0) = (u32x4){best[0], best[1], best[2], best[3]};
  unsigned lo = 0, hi = 0;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    lo |= (unsigned)bi[j] << (8 * j);
    hi |= (unsigned)bi[4 + j] << (8 * j);
  }
  *(uint2*)(idx + (size_t)o * Cp + c0) = make_uint2(lo, hi);
}

// gather form (deterministic): dy[n][h][w][c] = sum over the <= 4 windows that contain (h, w) and chose it, fp32, one rounding
__global__ __launch_bounds__(256) void cl16_maxpool_bwd_kernel(const unsigned short* __restrict__ dout,
                                                              const unsigned char* __restrict__ idx,
                                                              unsigned short* __restrict__ dy, int Cp, int H, int W, int Ho,
                                                              int Wo, unsigned total) {
  const unsigned i = blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const unsigned pc = Cp >> 3, piece = i % pc, q = i / pc;
  const int w = (int)(q % (unsigned)W), h = (int)((q / (unsigned)W) % (unsigned)H);
  const unsigned n = q / (unsigned)(W * H);
  const int c0 = piece * 8;
  float g[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (int ho = (h + 1) / 2 - 1; ho <= (h + 1) / 2; ++ho) {
    if (ho < 0 || ho >= Ho) continue;
    const int dh = h - (ho * 2 - 1);
    if (dh < 0 || dh > 2) continue;
    for (int wo = (w + 1) / 2 - 1; wo <= (w + 1) / 2; ++wo) {
      if (wo < 0 || wo >= Wo) continue;
      const int dw = w - (wo * 2 - 1);
      if (dw < 0 || dw > 2) continue;
      const size_t o = ((size_t)(n * Ho + ho) * Wo + wo) * Cp + c0;
      const u32x4 d = *(const u32x4*)(dout + o);
      const uint2 ix = *(const uint2*)(idx + o);
      const unsigned tap = (unsigned)(dh * 3 + dw);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const unsigned t = ((j < 4 ? ix.x : ix.y) >> (8 * (j & 3))) & 0xFFu;
        const float dv = (j & 1) ? bf_hi(d[j >> 1]) : bf_lo(d[j >> 1]);
        if (t == tap) g[j] += dv;
      }
    }
  }
  *(u32x4*)(dy + (size_t)q * Cp + c0) = (u32x4){pack_bf2(g[0], g[1]), pack_bf2(g[2], g[3]), pack_bf2(g[4], g[5]), pack_bf2(g[6], g[7])};
}


#ifndef SLV_EW_NT
#define SLV_EW_NT 1       // 1: the elementwise passes read / write their streams with the non-temporal hint (A/B: see r05 notes)
#endif
__device__ __forceinline__ u32x4 ew_ld(const unsigned short* p) {
#if SLV_EW_NT
  return __builtin_nontemporal_load((const u32x4*)p);
#else
  return *(const u32x4*)p;
#endif
}
__device__ __forceinline__ void ew_st(unsigned short* p, u32x4 v) {
#if SLV_EW_NT
  __builtin_nontemporal_store(v, (u32x4*)p);
#else
  *(u32x4*)p = v;
#endif
}
// ------------------------------------------------------------------------------------------ BatchNorm, channels last
// Elementwise kernels on [P][Cp] bf16: one thread per 16-byte piece (8 channels).  The grid stride (gridDim.x * 256) is a
// multiple of Cp/8 (cl16_ew_blocks), so a thread keeps ITS 8 channels for every position it visits and loads their
// per-channel coefficients once, into registers (the first version fetched them per element: 0.8 - 1.3 TB/s).
// out = relu?( x*s + h + (res ? (rss ? res*rs + rh : res) : 0) )
template <bool RES, bool RSS>
__global__ __launch_bounds__(256) void cl16_bn_act_kernel(const unsigned short* __restrict__ x, const float* __restrict__ ss,
                                                          const unsigned short* __restrict__ res,
                                                          const float* __restrict__ rss, int relu,
                                                          unsigned short* __restrict__ out, int C, int Cp,
                                                          unsigned total /* P * Cp / 8 */) {
  const unsigned pieces = Cp >> 3;
  const unsigned first = blockIdx.x * 256u + threadIdx.x, c0 = (first % pieces) * 8;
  float s[8], h[8], rs[8], rh[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const bool ok = c0 + i < (unsigned)C;
    s[i] = ok ? ss[c0 + i] : 0.f;
    h[i] = ok ? ss[C + c0 + i] : 0.f;
    rs[i] = (RSS && ok) ? rss[c0 + i] : 0.f;
    rh[i] = (RSS && ok) ? rss[C + c0 + i] : 0.f;
  }
  const bool tail = c0 + 8 > (unsigned)C;                        // this piece holds padding channels: keep them zero
  for (unsigned idx = first; idx < total; idx += gridDim.x * 256u) {
    const u32x4 xv = ew_ld(x + (size_t)idx * 8);
    u32x4 rv = {0u, 0u, 0u, 0u};
    if (RES) rv = ew_ld(res + (size_t)idx * 8);
    u32x4 o;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float v[2];
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        float t = bn_affine(e ? bf_hi(xv[i]) : bf_lo(xv[i]), s[2 * i + e], h[2 * i + e]);
        if (RES) {
          float r = e ? bf_hi(rv[i]) : bf_lo(rv[i]);
          if (RSS) {
            r = bn_affine(r, rs[2 * i + e], rh[2 * i + e]);
            // relu bit 1: the residual is the ACTIVATED output of its BatchNorm as a consumer's load prologue makes it
            // (ReLU, one rounding to bf16) -- the stem's un-materialised block output (engine.video_stage_forward)
            if (relu & 2) r = bf_lo(pack_bf2(fmaxf(r, 0.f), 0.f));
          }
          t += r;
        }
        if (relu & 1) t = fmaxf(t, 0.f);
        if (tail && c0 + 2 * i + e >= (unsigned)C) t = 0.f;
        v[e] = t;
      }
      o[i] = pack_bf2(v[0], v[1]);
    }
    ew_st(out + (size_t)idx * 8, o);
  }
}

// BatchNorm backward reductions on [P][Cp] bf16 (cf. bn_bwd_reduce_kernel of csrc/elementwise.hip):
//   g' = mask * g ; part[c][nsplit][2] = { sum g', sum g' * xhat(x) } (+ part2 with xhat2(x2) for the downsample BN)
// MASK 0: none; 1: own BN output > 0 (x*s + h); 2: block output v > 0 (g' is written to gout).
// One block per position slice: thread = (row r = tid / pieces, 16-byte piece); rows r, r + RPB, ... of the slice.
template <int MASK, bool TWO>
__global__ __launch_bounds__(256) void cl16_bn_bwd_reduce_kernel(const unsigned short* __restrict__ gin,
                                                                 const unsigned short* __restrict__ x,
                                                                 const float* __restrict__ mi, const float* __restrict__ ss,
                                                                 const unsigned short* __restrict__ v,
                                                                 const unsigned short* __restrict__ x2,
                                                                 const float* __restrict__ mi2,
                                                                 unsigned short* __restrict__ gout,
                                                                 float* __restrict__ part, float* __restrict__ part2,
                                                                 int C, int Cp, unsigned P, unsigned per, int nsplit) {
  extern __shared__ float sh[];                                  // [rpb][3][Cp]
  const int pieces = Cp >> 3, rpb = 256 / pieces > 0 ? 256 / pieces : 1;
  const int r = threadIdx.x / pieces, pc = threadIdx.x - r * pieces;
  const unsigned e0 = blockIdx.x * per, e1 = min(e0 + per, P);
  float a0[8], a1[8], a2[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) a0[i] = a1[i] = a2[i] = 0.f;
  if (r < rpb) {
    float mean[8], inv[8], mean2[8], inv2[8], s_[8], h_[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int c = pc * 8 + i;
      const bool okc = c < C;
      mean[i] = okc ? mi[c] : 0.f;
      inv[i] = okc ? mi[C + c] : 0.f;
      mean2[i] = (TWO && okc) ? mi2[c] : 0.f;
      inv2[i] = (TWO && okc) ? mi2[C + c] : 0.f;
      s_[i] = (MASK == 1 && okc) ? ss[c] : 0.f;
      h_[i] = (MASK == 1 && okc) ? ss[C + c] : 0.f;
    }
    for (unsigned p = e0 + r; p < e1; p += rpb) {
      const size_t ad = ((size_t)p * pieces + pc) * 8;
      u32x4 gv = ew_ld(gin + ad);
      const u32x4 xv = ew_ld(x + ad);
      u32x4 vv = {0u, 0u, 0u, 0u}, x2v = {0u, 0u, 0u, 0u};
      if (MASK == 2) vv = ew_ld(v + ad);
      if (TWO) x2v = ew_ld(x2 + ad);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const unsigned wsel = i >> 1;
        const bool hi = i & 1;
        float gg = hi ? bf_hi(gv[wsel]) : bf_lo(gv[wsel]);
        const float xx = hi ? bf_hi(xv[wsel]) : bf_lo(xv[wsel]);
        if (MASK == 1) gg = bn_affine(xx, s_[i], h_[i]) > 0.f ? gg : 0.f;
        if (MASK == 2) gg = (hi ? bf_hi(vv[wsel]) : bf_lo(vv[wsel])) > 0.f ? gg : 0.f;
        a0[i] += gg;
        a1[i] += gg * ((xx - mean[i]) * inv[i]);
        if (TWO) a2[i] += gg * (((hi ? bf_hi(x2v[wsel]) : bf_lo(x2v[wsel])) - mean2[i]) * inv2[i]);
        if (MASK == 2) {                                         // masked gradient: zero the element in place
          if (!(((hi ? bf_hi(vv[wsel]) : bf_lo(vv[wsel])) > 0.f))) gv[wsel] &= hi ? 0x0000FFFFu : 0xFFFF0000u;
        }
      }
      if (MASK == 2) ew_st(gout + ad, gv);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      sh[(r * 3 + 0) * Cp + pc * 8 + i] = a0[i];
      sh[(r * 3 + 1) * Cp + pc * 8 + i] = a1[i];
      sh[(r * 3 + 2) * Cp + pc * 8 + i] = a2[i];
    }
  }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += 256) {
    float t0 = 0.f, t1 = 0.f, t2 = 0.f;
    for (int rr = 0; rr < rpb; ++rr) {                           // fixed order
      t0 += sh[(rr * 3 + 0) * Cp + c];
      t1 += sh[(rr * 3 + 1) * Cp + c];
      t2 += sh[(rr * 3 + 2) * Cp + c];
    }
    part[((size_t)c * nsplit + blockIdx.x) * 2 + 0] = t0;
    part[((size_t)c * nsplit + blockIdx.x) * 2 + 1] = t1;
    if (TWO) {
      part2[((size_t)c * nsplit + blockIdx.x) * 2 + 0] = t0;
      part2[((size_t)c * nsplit + blockIdx.x) * 2 + 1] = t2;
    }
  }
}

// out = A1*mask*g + A2 + A3*x on [P][Cp] bf16 (b5 = s, h, A1, A2, A3 [5][C]); padding channels stay zero
__global__ __launch_bounds__(256) void cl16_bn_bwd_apply_kernel(const unsigned short* __restrict__ gin,
                                                                const unsigned short* __restrict__ x,
                                                                const float* __restrict__ b5, int relu,
                                                                unsigned short* __restrict__ out, int C, int Cp,
                                                                unsigned total) {
  const unsigned pieces = Cp >> 3;
  const unsigned first = blockIdx.x * 256u + threadIdx.x, c0 = (first % pieces) * 8;
  float s[8], h[8], a1[8], a2[8], a3[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const bool ok = c0 + i < (unsigned)C;
    s[i] = ok ? b5[c0 + i] : 0.f;
    h[i] = ok ? b5[C + c0 + i] : 0.f;
    a1[i] = ok ? b5[2 * C + c0 + i] : 0.f;
    a2[i] = ok ? b5[3 * C + c0 + i] : 0.f;
    a3[i] = ok ? b5[4 * C + c0 + i] : 0.f;
  }
  for (unsigned idx = first; idx < total; idx += gridDim.x * 256u) {
    const u32x4 gv = ew_ld(gin + (size_t)idx * 8);
    const u32x4 xv = ew_ld(x + (size_t)idx * 8);
    u32x4 o;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float v[2];
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int k = 2 * i + e;
        const float xx = e ? bf_hi(xv[i]) : bf_lo(xv[i]);
        float gg = e ? bf_hi(gv[i]) : bf_lo(gv[i]);
        if (relu && !(bn_affine(xx, s[k], h[k]) > 0.f)) gg = 0.f;
        v[e] = bn_bwd_apply1(gg, xx, a1[k], a2[k], a3[k]);       // padding channels: all coefficients are zero
      }
      o[i] = pack_bf2(v[0], v[1]);
    }
    ew_st(out + (size_t)idx * 8, o);
  }
}

// AdaptiveAvgPool(1) backward: dv[n][s][c] = dout[n][c] / S (bf16, padding channels zero)
__global__ __launch_bounds__(256) void cl16_avgpool_bwd_kernel(const float* __restrict__ dout, unsigned short* __restrict__ dv,
                                                               unsigned S, int C, int Cp, unsigned total) {
  const unsigned pieces = Cp >> 3;
  const float inv = 1.f / (float)S;
  for (unsigned idx = blockIdx.x * 256u + threadIdx.x; idx < total; idx += gridDim.x * 256u) {
    const unsigned pc = idx % pieces, n = (idx / pieces) / S;
    u32x4 o;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const unsigned c = pc * 8 + 2 * i;
      const float a = c < (unsigned)C ? dout[(size_t)n * C + c] * inv : 0.f;
      const float b = c + 1 < (unsigned)C ? dout[(size_t)n * C + c + 1] * inv : 0.f;
      o[i] = pack_bf2(a, b);
    }
    *(u32x4*)(dv + (size_t)idx * 8) = o;
  }
}

// bf16 [N][S][Cp] -> fp32 N,C,S (tests / inspection: the inverse of slv_to_cl16)
__global__ __launch_bounds__(256) void cl16_from_kernel(const unsigned short* __restrict__ x, float* __restrict__ y, int C,
                                                        int Cp, unsigned S, unsigned total /* N*C*S */) {
  const unsigned idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const unsigned s = idx % S, c = (idx / S) % C, n = idx / (S * (unsigned)C);
  y[idx] = bf2f(x[((size_t)n * S + s) * Cp + c]);
}

// grid of an elementwise launch: gridDim.x * 256 must be a multiple of Cp / 8 (see cl16_bn_act_kernel)
static unsigned cl16_ew_blocks(long long total, int Cp) {
  const int pieces = Cp / 8;
  int gcd = 256, b = pieces;
  while (b) { const int t = gcd % b; gcd = b; b = t; }
  const int unit = pieces / gcd;                                  // blocks must come in multiples of this
  long long want = (total + 2047) / 2048;                         // >= 8 pieces per thread: the per-thread coefficient loads
  if (want > 4096) want = 4096;                                   // (up to 10 x 16 B) must not outweigh the data
  long long blocks = (want + unit - 1) / unit * unit;
  return (unsigned)blocks;
}

template <int WM, int WN>
static int wgrad_launch(const ClWgrad& g, const void* dy, const void* x, const float* in_ss, float* part, hipStream_t st) {
  const FastDiv dWo = make_fastdiv(g.Wo), dHo = make_fastdiv(g.Ho), dTo = make_fastdiv(g.To);
  const unsigned blocks = (unsigned)g.mtiles * g.ntiles * g.kslices;
  if (in_ss)
    hipLaunchKernelGGL((cl16_wgrad_kernel<WM, WN, 1>), dim3(blocks), dim3(256), 0, st, (const unsigned short*)dy,
                       (const unsigned short*)x, in_ss, part, g, dWo, dHo, dTo);
  else
    hipLaunchKernelGGL((cl16_wgrad_kernel<WM, WN, 0>), dim3(blocks), dim3(256), 0, st, (const unsigned short*)dy,
                       (const unsigned short*)x, in_ss, part, g, dWo, dHo, dTo);
  return 0;
}

}  // namespace slv

extern "C" {

int slv_cl16_w_transform(const float* w, void* wf_bf16, void* wt_bf16, int Cout, int Cin, int taps, int Cin_p,
                         int Cout_p, int mrows_fwd, int mrows_dgrad, int patch_kw, slv_stream_t stream) {
  using namespace slv;
  SLV_CHECK_ARG(w && (wf_bf16 || wt_bf16) && Cout > 0 && Cin > 0 && taps > 0, "bad argument");
  SLV_CHECK_ARG((Cin_p & 31) == 0 && (Cout_p & 31) == 0 && Cin_p >= Cin && Cout_p >= Cout, "padded channel counts");
  SLV_CHECK_ARG(!patch_kw || (Cin_p == 32 && patch_kw * Cin <= 32 && !wt_bf16), "patch layout: forward only, kw*Cin <= 32");
  const long long nf = wf_bf16 ? (long long)taps * (Cin_p / 32) * mrows_fwd * 32 : 0;
  const long long nt = wt_bf16 ? (long long)taps * (Cout_p / 32) * mrows_dgrad * 32 : 0;
  SLV_CHECK_ARG(nf + nt < 0xFFFFFF00LL, "weights too large");
  SLV_CHECK_ARG((!wf_bf16 || mrows_fwd >= Cout) && (!wt_bf16 || mrows_dgrad >= Cin), "layout rows");
  hipLaunchKernelGGL(cl16_w_transform_kernel, dim3((unsigned)((nf + nt + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w,
                     (unsigned short*)wf_bf16, (unsigned short*)wt_bf16, Cout, Cin, taps, Cin_p, Cout_p, mrows_fwd,
                     mrows_dgrad, patch_kw, (unsigned)nf, (unsigned)nt);
  SLV_LAUNCH_CHECK();
  return 0;
}

int slv_cl16_w_transform_jobs(const int32_t* jobs_dev, int32_t njobs, int32_t blocks_per_job, slv_stream_t stream) {
  using namespace slv;
  SLV_CHECK_ARG(jobs_dev && njobs > 0 && njobs <= 65535 && blocks_per_job > 0 && blocks_per_job <= 4096, "bad argument");
  hipLaunchKernelGGL(cl16_w_transform_jobs_kernel, dim3((unsigned)blocks_per_job, (unsigned)njobs), dim3(256), 0, (hipStream_t)stream,
                     (const ClWJob*)jobs_dev);
  SLV_LAUNCH_CHECK();
  return 0;
}

int32_t slv_cl16_wgrad_words(void) { return slv::CLW_WORDS; }

size_t slv_cl16_wgrad_ws_bytes(const int32_t* clw, int wm, int wn) {
  slv::ClWgrad g;
  memcpy(&g, clw, sizeof(g));
  size_t general = (size_t)g.kslices * g.mtiles * wm * 32 * g.ntiles * wn * 32 * sizeof(float);
  if (slv::wgrad_acc_applies(g.N, g.Ti, g.Hi, g.Wi, g.Cin_p, g.Cin, g.Cout_p, 144, g.kt, g.kh, g.kw, g.st, g.sh, g.sw, g.pt, g.ph,
                             g.pw, g.To, g.Ho, g.Wo) && slv::wgrad_acc_ws_bytes() > general)
    general = slv::wgrad_acc_ws_bytes();
  slv::ClWgrad3 g3;
  int wm3;
  if (slv::wgrad3_plan(g.N, g.Ti, g.Hi, g.Wi, g.Cin_p, g.Cin, g.Cout_p, g.kt, g.kh, g.kw, g.st, g.sh, g.sw, g.pt, g.ph, g.pw,
                       g.To, g.Ho, g.Wo, &wm3, &g3)) {
    const size_t patch = slv::wgrad3_ws_bytes(g3, wm3);
    return patch > general ? patch : general;
  }
  slv::ClWgradT gt;
  int wmt, nct;
  if (slv::wgrad_t_plan(g.N, g.Ti, g.Hi, g.Wi, g.Cin_p, g.Cin, g.Cout_p, g.kt, g.kh, g.kw, g.st, g.sh, g.sw, g.pt, g.ph, g.pw,
                        g.To, g.Ho, g.Wo, &wmt, &nct, &gt)) {
    const size_t col = slv::wgrad_t_ws_bytes(gt, wmt);
    return col > general ? col : general;
  }
  return general;
}

int slv_cl16_wgrad(const int32_t* clw, int wm, int wn, const void* dy_bf16, const void* x_bf16,
                   const float* in_scale_shift, float* dw, int Cout, int patch_kw, void* ws, size_t ws_bytes,
                   slv_stream_t stream) {
  using namespace slv;
  SLV_CHECK_ARG(clw && dy_bf16 && x_bf16 && dw && ws, "null pointer");
  ClWgrad g;
  memcpy(&g, clw, sizeof(g));
  SLV_CHECK_ARG(wm >= 2 && wm <= 5 && wn >= 2 && wn <= 5, "tile (wm, wn in 2..5: 64..160 rows / columns)");
  SLV_CHECK_ARG(g.N > 0 && (g.Cin_p & 31) == 0 && (g.Cout_p & 31) == 0 && g.Cin > 0 && g.Cin <= g.Cin_p && Cout > 0 &&
                    Cout <= g.Cout_p, "channel counts");
  SLV_CHECK_ARG(g.kt > 0 && g.kh > 0 && g.kw > 0 && g.Ncols == g.kt * g.kh * g.kw * g.Cin_p, "columns");
  SLV_CHECK_ARG(g.mtiles * wm * 32 >= g.Cout_p && g.ntiles * wn * 32 >= g.Ncols && g.mtiles > 0 && g.ntiles > 0, "tiles");
  SLV_CHECK_ARG(g.kslices > 0 && g.kper > 0 && (g.kper & 31) == 0 &&
                    (long long)g.kslices * g.kper >= (long long)g.N * g.To * g.Ho * g.Wo, "K slices");
  SLV_CHECK_ARG((long long)g.N * g.Ti * g.Hi * g.Wi * g.Cin_p * 2 < 0xFFFFFFF0LL &&
                    (long long)g.N * g.To * g.Ho * g.Wo * g.Cout_p * 2 < 0xFFFFFFF0LL, "tensor beyond the 32-bit buffer range");
  SLV_CHECK_ARG(!in_scale_shift || g.Cin_p <= 1152, "prologue table: Cin_p <= 1152");
  SLV_CHECK_ARG(!patch_kw || (g.Cin_p == 32 && g.kt == 1 && g.kw == 1), "patch layout");
  SLV_CHECK_ARG(ws_bytes >= slv_cl16_wgrad_ws_bytes(clw, wm, wn), "workspace too small");
  hipStream_t st = (hipStream_t)stream;
  float* part = (float*)ws;
  const int taps = g.kt * g.kh * g.kw;
  if (!patch_kw && wgrad_acc_applies(g.N, g.Ti, g.Hi, g.Wi, g.Cin_p, g.Cin, g.Cout_p, Cout, g.kt, g.kh, g.kw, g.st, g.sh, g.sw,
                                     g.pt, g.ph, g.pw, g.To, g.Ho, g.Wo)) {   // layer-1 spatial: dW resident in accumulators
    const int rc = wgrad_acc_launch(g.N, g.Ti, g.Hi, g.Wi, dy_bf16, x_bf16, in_scale_shift, part, st);
    if (rc) return rc;
    const size_t ldpa = (size_t)g.Ncols;
    hipLaunchKernelGGL(cl16_wgrad_reduce_kernel, dim3((g.Ncols + 255) / 256, Cout), dim3(256), 0, st, part, dw, Cout, g.Cin,
                       taps, g.Cin_p, wgrad_acc_blocks(), (size_t)Cout * ldpa, ldpa, 0, (unsigned)g.Ncols);
    SLV_LAUNCH_CHECK();
    return 0;
  }
  {                                                                // stride-1 (1,3,3): the rolling-patch kernel
    ClWgrad3 g3;
    int wm3;
    if (!patch_kw && wgrad3_plan(g.N, g.Ti, g.Hi, g.Wi, g.Cin_p, g.Cin, g.Cout_p, g.kt, g.kh, g.kw, g.st, g.sh, g.sw, g.pt,
                                 g.ph, g.pw, g.To, g.Ho, g.Wo, &wm3, &g3)) {
      wgrad3_launch(g3, wm3, dy_bf16, x_bf16, in_scale_shift, part, st);
      SLV_LAUNCH_CHECK();
      const size_t ldp3 = (size_t)g.Ncols;
      hipLaunchKernelGGL(cl16_wgrad_reduce_kernel, dim3((g.Ncols + 255) / 256, Cout), dim3(256), 0, st, part, dw, Cout, g.Cin,
                         taps, g.Cin_p, g3.kslices, (size_t)g3.mtiles * wm3 * 32 * ldp3, ldp3, 0, (unsigned)g.Ncols);
      SLV_LAUNCH_CHECK();
      return 0;
    }
  }
  {                                                                // stride-1 (3,1,1): the column-order kernel
    ClWgradT gt;
    int wmt, nct;
    if (!patch_kw && wgrad_t_plan(g.N, g.Ti, g.Hi, g.Wi, g.Cin_p, g.Cin, g.Cout_p, g.kt, g.kh, g.kw, g.st, g.sh, g.sw, g.pt,
                                  g.ph, g.pw, g.To, g.Ho, g.Wo, &wmt, &nct, &gt)) {
      wgrad_t_launch(gt, wmt, nct, dy_bf16, x_bf16, in_scale_shift, part, st);
      SLV_LAUNCH_CHECK();
      const size_t ldpt = (size_t)g.Ncols;
      hipLaunchKernelGGL(cl16_wgrad_reduce_kernel, dim3((g.Ncols + 255) / 256, Cout), dim3(256), 0, st, part, dw, Cout, g.Cin,
                         taps, g.Cin_p, gt.kslices, (size_t)gt.mtiles * wmt * 32 * ldpt, ldpt, 0, (unsigned)g.Ncols);
      SLV_LAUNCH_CHECK();
      return 0;
    }
  }
#define SLV_WG(A_, B_) \
  if (wm == A_ && wn == B_) wgrad_launch<A_, B_>(g, dy_bf16, x_bf16, in_scale_shift, part, st)
  SLV_WG(2, 2); SLV_WG(2, 3); SLV_WG(2, 4); SLV_WG(2, 5);
  SLV_WG(3, 2); SLV_WG(3, 3); SLV_WG(3, 4); SLV_WG(3, 5);
  SLV_WG(4, 2); SLV_WG(4, 3); SLV_WG(4, 4); SLV_WG(4, 5);
  SLV_WG(5, 2); SLV_WG(5, 3); SLV_WG(5, 4); SLV_WG(5, 5);
#undef SLV_WG
  SLV_LAUNCH_CHECK();
  const int Cin_w = patch_kw ? g.Cin / patch_kw : g.Cin;        // patch mode: g.Cin = kw * C patch channels in use
  const size_t ldp = (size_t)g.ntiles * wn * 32;
  hipLaunchKernelGGL(cl16_wgrad_reduce_kernel, dim3((g.Ncols + 255) / 256, Cout), dim3(256), 0, st, part, dw, Cout, Cin_w,
                     taps, g.Cin_p, g.kslices, (size_t)g.mtiles * wm * 32 * ldp, ldp, patch_kw, (unsigned)g.Ncols);
  SLV_LAUNCH_CHECK();
  return 0;
}

size_t slv_cl16_wgrad_bnr_ws_bytes(const int32_t* clw) {
  slv::ClWgrad g;
  memcpy(&g, clw, sizeof(g));
  slv::ClWgradT gt;
  int wmt, nct;
  if (!slv::wgrad_t2_plan(g.N, g.Ti, g.Hi, g.Wi, g.Cin_p, g.Cin, g.Cout_p, g.kt, g.kh, g.kw, g.st, g.sh, g.sw, g.pt, g.ph, g.pw,
                          g.To, g.Ho, g.Wo, &wmt, &nct, &gt))
    return 0;
  return slv::wgrad_t2_ws_bytes(gt, wmt);
}

int slv_cl16_wgrad_bnr(const int32_t* clw, const void* dy_bf16, const void* x_bf16, const float* in_scale_shift,
                       const float* in_mean_invstd, const float* w, float* dw, float* bn_part, int Cout, void* ws,
                       size_t ws_bytes, slv_stream_t stream) {
  using namespace slv;
  SLV_CHECK_ARG(clw && dy_bf16 && x_bf16 && in_scale_shift && in_mean_invstd && w && dw && bn_part && ws, "null pointer");
  ClWgrad g;
  memcpy(&g, clw, sizeof(g));
  ClWgradT gt;
  int wmt, nct;
  SLV_CHECK_ARG(wgrad_t2_plan(g.N, g.Ti, g.Hi, g.Wi, g.Cin_p, g.Cin, g.Cout_p, g.kt, g.kh, g.kw, g.st, g.sh, g.sw, g.pt, g.ph,
                              g.pw, g.To, g.Ho, g.Wo, &wmt, &nct, &gt), "not a stride-1 (3,1,1) conv this kernel takes");
  SLV_CHECK_ARG(Cout > 0 && Cout <= g.Cout_p && ws_bytes >= wgrad_t2_ws_bytes(gt, wmt), "Cout / workspace");
  SLV_CHECK_ARG((long long)g.N * g.Ti * g.Hi * g.Wi * g.Cin_p * 2 < 0xFFFFFFF0LL &&
                    (long long)g.N * g.To * g.Ho * g.Wo * g.Cout_p * 2 < 0xFFFFFFF0LL, "tensor beyond the 32-bit buffer range");
  return wgrad_t2_launch(gt, wmt, nct, dy_bf16, x_bf16, in_scale_shift, in_mean_invstd, w, dw, bn_part, Cout, (float*)ws,
                         (hipStream_t)stream);
}

int slv_cl16_bn_act(const void* x_bf16, const float* scale_shift, const void* res_bf16, const float* res_scale_shift,
                    int relu, void* out_bf16, int64_t P, int C, int Cp, slv_stream_t stream) {
  using namespace slv;
  SLV_CHECK_ARG(x_bf16 && scale_shift && out_bf16 && P > 0 && C > 0 && Cp >= C && (Cp & 7) == 0, "bad argument");
  const long long total = P * (Cp / 8);
  SLV_CHECK_ARG(total < 0xFFFFFF00LL, "tensor too large");
  SLV_CHECK_ARG(!res_scale_shift || res_bf16, "res_scale_shift without res");
  const unsigned blocks = cl16_ew_blocks(total, Cp);
#define SLV_ACT(R_, S_)                                                                                                  \
  hipLaunchKernelGGL((cl16_bn_act_kernel<R_, S_>), dim3(blocks), dim3(256), 0, (hipStream_t)stream,                      \
                     (const unsigned short*)x_bf16, scale_shift, (const unsigned short*)res_bf16, res_scale_shift, relu, \
                     (unsigned short*)out_bf16, C, Cp, (unsigned)total)
  if (!res_bf16) SLV_ACT(false, false);
  else if (!res_scale_shift) SLV_ACT(true, false);
  else SLV_ACT(true, true);
#undef SLV_ACT
  SLV_LAUNCH_CHECK();
  return 0;
}

int32_t slv_cl16_bn_bwd_nsplit(int64_t P, int Cp) {
  const int pieces = Cp / 8, rpb = 256 / pieces > 0 ? 256 / pieces : 1;
  // ~32 rows per thread at least, at most 1024 slices
  long long ns = (P + (long long)rpb * 32 - 1) / ((long long)rpb * 32);
  if (ns > 1024) ns = 1024;
  if (ns < 1) ns = 1;
  return (int32_t)ns;
}

int slv_cl16_bn_bwd_reduce(const void* g_bf16, const void* x_bf16, const float* mean_invstd, const float* scale_shift_mask,
                           const void* v_mask_bf16, const void* x2_bf16, const float* mean_invstd2, void* g_out_bf16,
                           float* partial, float* partial2, int64_t P, int C, int Cp, int nsplit, slv_stream_t stream) {
  using namespace slv;
  SLV_CHECK_ARG(g_bf16 && x_bf16 && mean_invstd && partial && P > 0 && C > 0 && Cp >= C && (Cp & 7) == 0 && nsplit > 0,
                "bad argument");
  SLV_CHECK_ARG(!(scale_shift_mask && v_mask_bf16), "one mask at most");
  SLV_CHECK_ARG(!v_mask_bf16 || g_out_bf16, "the masked gradient needs an output");
  SLV_CHECK_ARG(!x2_bf16 || (mean_invstd2 && partial2), "second BatchNorm: mean_invstd2, partial2");
  SLV_CHECK_ARG(P < 0xFFFFFF00LL && Cp <= 2048, "tensor too large");
  const int pieces = Cp / 8, rpb = 256 / pieces > 0 ? 256 / pieces : 1;
  SLV_CHECK_ARG(pieces <= 256, "Cp <= 2048");
  const unsigned per = (unsigned)((P + nsplit - 1) / nsplit);
  const size_t shb = (size_t)rpb * 3 * Cp * sizeof(float);
  const int mask = scale_shift_mask ? 1 : (v_mask_bf16 ? 2 : 0);
  const bool two = x2_bf16 != nullptr;
#define SLV_R(M_, T_)                                                                                               \
  hipLaunchKernelGGL((cl16_bn_bwd_reduce_kernel<M_, T_>), dim3(nsplit), dim3(256), shb, (hipStream_t)stream,        \
                     (const unsigned short*)g_bf16, (const unsigned short*)x_bf16, mean_invstd, scale_shift_mask,    \
                     (const unsigned short*)v_mask_bf16, (const unsigned short*)x2_bf16, mean_invstd2,               \
                     (unsigned short*)g_out_bf16, partial, partial2, C, Cp, (unsigned)P, per, nsplit)
  if (mask == 0 && !two) SLV_R(0, false);
  else if (mask == 0) SLV_R(0, true);
  else if (mask == 1 && !two) SLV_R(1, false);
  else if (mask == 1) SLV_R(1, true);
  else if (!two) SLV_R(2, false);
  else SLV_R(2, true);
#undef SLV_R
  SLV_LAUNCH_CHECK();
  return 0;
}

int slv_cl16_bn_bwd_apply(const void* g_bf16, const void* x_bf16, const float* bwd5, int relu, void* out_bf16, int64_t P,
                          int C, int Cp, slv_stream_t stream) {
  using namespace slv;
  SLV_CHECK_ARG(g_bf16 && x_bf16 && bwd5 && out_bf16 && P > 0 && C > 0 && Cp >= C && (Cp & 7) == 0, "bad argument");
  const long long total = P * (Cp / 8);
  SLV_CHECK_ARG(total < 0xFFFFFF00LL, "tensor too large");
  const unsigned blocks = cl16_ew_blocks(total, Cp);
  hipLaunchKernelGGL(cl16_bn_bwd_apply_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream,
                     (const unsigned short*)g_bf16, (const unsigned short*)x_bf16, bwd5, relu, (unsigned short*)out_bf16, C,
                     Cp, (unsigned)total);
  SLV_LAUNCH_CHECK();
  return 0;
}

int slv_cl16_avgpool_bwd(const float* dout, void* dv_bf16, int64_t N, int64_t S, int C, int Cp, slv_stream_t stream) {
  using namespace slv;
  SLV_CHECK_ARG(dout && dv_bf16 && N > 0 && S > 0 && C > 0 && Cp >= C && (Cp & 7) == 0, "bad argument");
  const long long total = N * S * (Cp / 8);
  SLV_CHECK_ARG(total < 0xFFFFFF00LL && S < 0x7FFFFFFFLL, "tensor too large");
  const unsigned blocks = (unsigned)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
  hipLaunchKernelGGL(cl16_avgpool_bwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, dout,
                     (unsigned short*)dv_bf16, (unsigned)S, C, Cp, (unsigned)total);
  SLV_LAUNCH_CHECK();
  return 0;
}

int slv_cl16_bnrelu_maxpool_fwd(const void* x_bf16, const float* scale_shift, void* out_bf16, uint8_t* idx, int64_t N, int C,
                                int Cp, int H, int W, slv_stream_t stream) {
  using namespace slv;
  SLV_CHECK_ARG(x_bf16 && scale_shift && out_bf16 && idx && N > 0 && C > 0 && Cp >= C && (Cp & 7) == 0 && H > 0 && W > 0,
                "bad argument");
  const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
  const long long total = N * Ho * Wo * (Cp / 8);
  SLV_CHECK_ARG(total < 0xFFFFFF00LL && N * H * W * (long long)Cp < 0xFFFFFF00LL, "tensor too large");
  hipLaunchKernelGGL(cl16_bnrelu_maxpool_fwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     (const unsigned short*)x_bf16, scale_shift, (unsigned short*)out_bf16, idx, C, Cp, H, W, Ho, Wo,
                     (unsigned)total);
  SLV_LAUNCH_CHECK();
  return 0;
}

int slv_cl16_maxpool_bwd(const void* dout_bf16, const uint8_t* idx, void* dy_bf16, int64_t N, int Cp, int H, int W,
                         slv_stream_t stream) {
  using namespace slv;
  SLV_CHECK_ARG(dout_bf16 && idx && dy_bf16 && N > 0 && Cp > 0 && (Cp & 7) == 0 && H > 0 && W > 0, "bad argument");
  const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
  const long long total = N * H * W * (Cp / 8);
  SLV_CHECK_ARG(total < 0xFFFFFF00LL, "tensor too large");
  hipLaunchKernelGGL(cl16_maxpool_bwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     (const unsigned short*)dout_bf16, idx, (unsigned short*)dy_bf16, Cp, H, W, Ho, Wo, (unsigned)total);
  SLV_LAUNCH_CHECK();
  return 0;
}

int slv_from_cl16(const void* x_bf16, float* y, int64_t N, int C, int Cp, int64_t S, slv_stream_t stream) {
  using namespace slv;
  SLV_CHECK_ARG(x_bf16 && y && N > 0 && C > 0 && Cp >= C && S > 0, "bad argument");
  const long long total = N * C * S;
  SLV_CHECK_ARG(total < 0xFFFFFF00LL, "tensor too large");
  hipLaunchKernelGGL(cl16_from_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     (const unsigned short*)x_bf16, y, C, Cp, (unsigned)S, (unsigned)total);
  SLV_LAUNCH_CHECK();
  return 0;
}

}  // extern "C"
