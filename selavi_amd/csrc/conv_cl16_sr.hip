// 16-bit MFMA path: the layer-1 spatial convolution Conv3d(64 -> 144, (1,3,3), stride 1, "same") -- the heaviest launch
// of the trunk (a fifth of the forward's bytes, bench.py's cfg5 roofline kernel) -- with the WEIGHTS RESIDENT IN
// REGISTERS and the data movement on a wave of its own.
//
// By arithmetic intensity (128 FLOP/B) this conv is bound by HBM; the tile kernel (csrc/conv_cl16_s3.hip) spends its
// time re-streaming the 166 KB weight tensor through LDS for every 128 positions (LDS-DMA issue + a barrier per tap),
// on per-tap border masks, and on prologue / epilogue phases that nothing overlaps.  Here:
//   * one workgroup of 4 waves per CU, persistent over 8 x 8-pixel tiles of one frame;
//   * waves 0-2 are MFMA waves: wave w owns output channels 48 w .. 48 w + 47 and holds their whole weight slice --
//     9 taps x 2 chunks x 3 tiles = 54 A fragments = 216 registers, in the accumulator half of the register file
//     (asm MFMAs with an "a" operand) -- for the kernel's lifetime.  Per tile: 72 ds_read_b128 of the input patch
//     (one base register, every tap / chunk / fragment an immediate offset), 216 MFMAs, the bf16 tile into LDS;
//   * wave 3 is the data-movement wave: it fetches the 10 x 10-pixel patch of the NEXT tile (16-byte pieces, each lane
//     a fixed channel piece, so BatchNorm + ReLU coefficients stay in registers), applies BatchNorm + ReLU once per
//     element, writes zeros for the halo outside the image (no per-tap masks anywhere), stores the PREVIOUS tile's
//     output in whole 320-byte channel rows, and takes the BatchNorm statistics of that tile on its own matrix core;
//   * patch and output tile are double buffered in LDS (one workgroup per CU: 92 KB of the 160), ONE barrier per tile.
// LDS patch: pixel (py, px) of the 10 x 10 patch at ((py * 16 + px) * 160) bytes -- pitch 16, rows of 128 + 32 bytes:
// found by enumeration to be conflict-free for ds_read_b128 fragments whose 16 positions are 2 tile rows x 8 columns,
// for every tap shift, without any XOR (so taps stay immediates).
#include "cl16.hpp"
#include "../../include/selavi_hip.h"

namespace slv {

constexpr int SR_T = 8;                               // tile edge (pixels)
constexpr int SR_PW = SR_T + 2;                       // patch edge
constexpr int SR_PITCH = 16, SR_ROWB = 160;           // LDS patch geometry (see above)
constexpr int SR_PATCH = SR_PW * SR_PITCH * SR_ROWB;  // 25 600 B
constexpr int SR_CIN = 64, SR_COUT = 144, SR_COUTP = 160;
constexpr int SR_OROW = SR_COUTP * 2 + 16;            // bytes per pixel row of the output stage
constexpr int SR_OST = SR_T * SR_T * SR_OROW;         // 21 504 B
constexpr int SR_LDS = 2 * SR_PATCH + 2 * SR_OST;
constexpr int SR_NIT = (SR_PW * SR_PW + 7) / 8;       // 13 load instructions cover the patch (8 rows of 8 pieces each)

__device__ __forceinline__ void sr_mfma(f32x4& acc, const bf16x8& a, const bf16x8& b) {
  asm("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc) : "a"(a), "v"(b));
}
__device__ __forceinline__ void sr_mfma0(f32x4& acc, const bf16x8& a, const bf16x8& b) {
  asm("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=v"(acc) : "a"(a), "v"(b));
}
__device__ __forceinline__ void sr_barrier() {        // LDS traffic of this wave complete, then the workgroup barrier;
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // global loads / stores stay in flight across it
  __builtin_amdgcn_s_barrier();
}

// PRO 1: rows are read as relu(x * s + h).  EPI 1: per-channel sum / sum of squares of the rounded outputs, one partial
// per workgroup: stat_sum / stat_sq [Cout][gridDim.x].
template <int PRO, int EPI>
__global__ __launch_bounds__(256, 1) void conv_cl16_sr_kernel(const unsigned short* __restrict__ x,
                                                             const unsigned short* __restrict__ wl,
                                                             unsigned short* __restrict__ y,
                                                             const float* __restrict__ in_ss,
                                                             float* __restrict__ stat_sum, float* __restrict__ stat_sq,
                                                             ClConv g, int ntiles, int th, int tw) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  unsigned char* const patch = lds;                   // [2][SR_PATCH]
  unsigned char* const ost = lds + 2 * SR_PATCH;      // [2][64][SR_OROW]
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int fr = lane & 15, fk = lane >> 4;
  const int nt = blockIdx.x < (unsigned)ntiles ? (ntiles - 1 - (int)blockIdx.x) / (int)gridDim.x + 1 : 0;

  if (wave < 3) {
    // ======================================================================== MFMA waves
    bf16x8 A[9][2][3];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int i = 0; i < 3; ++i)
          A[t][c][i] = *(const bf16x8*)(wl + ((size_t)((t * 2 + c) * SR_COUT + (wave * 3 + i) * 16 + fr) * 32 + fk * 8));
    // lane part of a fragment read: position fr of fragment nn is tile pixel (2 nn + (fr >> 3), fr & 7)
    const int lbase = (((fr >> 3) * SR_PITCH + (fr & 7)) * SR_ROWB) + fk * 16;
    const int obase = fr * SR_OROW + ((wave * 3) * 16 + fk * 4) * 2;
    sr_barrier();                                     // patch 0 is in LDS
    for (int n = 0; n < nt; ++n) {
      const unsigned char* src = patch + (n & 1) * SR_PATCH + lbase;
      f32x4 acc[3][4];
#pragma unroll
      for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          bf16x8 b[4];
#pragma unroll
          for (int nn = 0; nn < 4; ++nn)
            b[nn] = *(const bf16x8*)(src + ((2 * nn + t / 3) * SR_PITCH + (t % 3)) * SR_ROWB + c * 64);
#pragma unroll
          for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int nn = 0; nn < 4; ++nn) {
              if (t == 0 && c == 0) sr_mfma0(acc[i][nn], A[t][c][i], b[nn]);
              else sr_mfma(acc[i][nn], A[t][c][i], b[nn]);
            }
        }
      asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");      // 8-pass XDL result -> VALU read
      unsigned char* dst = ost + (n & 1) * SR_OST + obase;
#pragma unroll
      for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int nn = 0; nn < 4; ++nn) {
          const unsigned lo = pack_bf2(acc[i][nn][0], acc[i][nn][1]), hi = pack_bf2(acc[i][nn][2], acc[i][nn][3]);
          *(uint2*)(dst + nn * 16 * SR_OROW + i * 32) = make_uint2(lo, hi);
        }
      sr_barrier();
    }
    return;
  }

  // ========================================================================== data-movement wave
  const int T = g.Ti, H = g.Hi, W = g.Wi;
  const unsigned Ptot = (unsigned)g.N * T * H * W;
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, (int)(Ptot * (SR_CIN * 2u)), 0x00020000);
  const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc((void*)y, 0, (int)(Ptot * (SR_COUTP * 2u)), 0x00020000);
  // zero both output stages once: the padding channels 144..159 of every row stay zero
  for (int i = lane * 16; i < 2 * SR_OST; i += 64 * 16) *(u32x4*)(ost + i) = (u32x4){0u, 0u, 0u, 0u};
  // this lane's pieces of a patch: piece = lane & 7 of patch rows lr, lr + 8, ...
  const int piece = lane & 7, lr = lane >> 3;
  float ps[8], ph[8];
  if constexpr (PRO == 1) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int c = piece * 8 + e;
      ps[e] = c < g.Cin ? in_ss[c] : 0.f;
      ph[e] = c < g.Cin ? in_ss[g.Cin + c] : 0.f;
    }
  }
  int pyx[SR_NIT];                                    // py | px << 8 | live << 16 of the lane's patch rows
#pragma unroll
  for (int i = 0; i < SR_NIT; ++i) {
    const int r = lr + 8 * i, py = r / SR_PW, px = r - py * SR_PW;
    pyx[i] = py | (px << 8) | ((r < SR_PW * SR_PW) << 16);
  }
  struct Tile {
    int y0, x0;
    unsigned fpos;                                    // position of the frame's pixel (0, 0)
  };
  auto tile_of = [&](int k) __attribute__((always_inline)) {
    Tile t;
    const int id = blockIdx.x + k * gridDim.x;
    const int per = th * tw, f = id / per, rem = id - f * per, ty = rem / tw;
    t.y0 = ty * SR_T;
    t.x0 = (rem - ty * tw) * SR_T;
    t.fpos = (unsigned)f * H * W;
    if (k >= nt) t.y0 = 1 << 20;                      // (past the last tile: every row fails its bounds test, per lane)
    return t;
  };
  u32x4 st[SR_NIT];
  unsigned stv = 0;
  auto load_patch = [&](const Tile& t) __attribute__((always_inline)) {
    stv = 0;
#pragma unroll
    for (int i = 0; i < SR_NIT; ++i) {
      const int py = pyx[i] & 255, px = (pyx[i] >> 8) & 255;
      const int yy = t.y0 - 1 + py, xx = t.x0 - 1 + px;
      const bool ok = (pyx[i] >> 16) && (unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W;
      stv |= (unsigned)ok << i;
      const unsigned off = (t.fpos + (unsigned)(yy * W + xx)) * (SR_CIN * 2u) + piece * 16u;
      st[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, ok ? off : 0xFFFFFFF0u, 0, 0));
    }
  };
  auto store_patch = [&](int buf) __attribute__((always_inline)) {
    unsigned char* dst = patch + buf * SR_PATCH + piece * 16;
#pragma unroll
    for (int i = 0; i < SR_NIT; ++i) {
      const int py = pyx[i] & 255, px = (pyx[i] >> 8) & 255;
      u32x4 v = st[i];
      if constexpr (PRO == 1) {                       // zero padding AFTER the affine: the halo outside the image is zero
        const u32x4 a = affine_relu8(v, ps, ph);
        v = ((stv >> i) & 1) ? a : (u32x4){0u, 0u, 0u, 0u};
      }
      // (rows past the patch, i == SR_NIT - 1 only, land in the unused pitch columns 10..15 of patch row 9 + ...: keep them inside)
      const int row = (pyx[i] >> 16) ? py * SR_PITCH + px : SR_PW * SR_PITCH - 1;
      *(u32x4*)(dst + row * SR_ROWB) = v;
    }
  };
  float accS[9], accQ[9];
#pragma unroll
  for (int i = 0; i < 9; ++i) accS[i] = accQ[i] = 0.f;
  // output rows: 20 pieces of 16 bytes; 60 lanes cover 3 pixels per instruction
  const int opiece = lane % 20, olr = lane / 20;
  auto store_out = [&](const Tile& t, int buf) __attribute__((always_inline)) {
    const unsigned char* src = ost + buf * SR_OST;
    if constexpr (EPI == 1) {
      if (t.y0 + SR_T > H || t.x0 + SR_T > W) {       // ragged tile: pixels outside the image count as zero
        for (int i = lane; i < SR_T * SR_T * (SR_COUT / 8); i += 64) {
          const int p = i / (SR_COUT / 8), q = i - p * (SR_COUT / 8);
          if (t.y0 + (p >> 3) >= H || t.x0 + (p & 7) >= W) *(u32x4*)(ost + buf * SR_OST + p * SR_OROW + q * 16) = (u32x4){0u, 0u, 0u, 0u};
        }
      }
      wave_rows32_stats_acc<9>(src, SR_OROW, lane, accS, accQ);
      wave_rows32_stats_acc<9>(src + 32 * SR_OROW, SR_OROW, lane, accS, accQ);
    }
#pragma unroll 2
    for (int p0 = 0; p0 < SR_T * SR_T; p0 += 3) {     // (out-of-range offsets drop the store: no branches)
      const int p = p0 + olr;
      const int yy = t.y0 + (p >> 3), xx = t.x0 + (p & 7);
      const bool ok = lane < 60 && p < SR_T * SR_T && yy < H && xx < W;
      const u32x4 v = *(const u32x4*)(src + (p < SR_T * SR_T ? p : 0) * SR_OROW + opiece * 16);
      const unsigned off = (t.fpos + (unsigned)(yy * W + xx)) * (SR_COUTP * 2u) + opiece * 16u;
      __builtin_amdgcn_raw_buffer_store_b128(v, ry, ok ? off : 0xFFFFFFF0u, 0, 0);
    }
  };
  // ---- pipeline head: patch 0 into buffer 0, patch 1 requested
  load_patch(tile_of(0));
  store_patch(0);
  load_patch(tile_of(1));
  sr_barrier();
  for (int n = 0; n < nt; ++n) {
    store_patch((n + 1) & 1);                         // patch n + 1 (requested a step ago) -> the buffer tile n - 1 used
    load_patch(tile_of(n + 2));                       // patch n + 2: in flight across the barrier
    if (n >= 1) store_out(tile_of(n - 1), (n - 1) & 1);      // tile n - 1: statistics + 320-byte rows to memory
    sr_barrier();
  }
  if (nt > 0) store_out(tile_of(nt - 1), (nt - 1) & 1);
  if constexpr (EPI == 1) {
#pragma unroll
    for (int i = 0; i < 9; ++i) {
      const int c = i * 16 + fr;
      if (fk == 0) stat_sum[(size_t)c * gridDim.x + blockIdx.x] = accS[i];
      if (fk == (fr >> 2)) stat_sq[(size_t)c * gridDim.x + blockIdx.x] = accQ[i];
    }
  }
}

static bool sr_enabled() {
  static const bool enabled = []() {
    const char* e = getenv("SELAVI_CL16_SR");
    return !(e && e[0] == '0');
  }();
  return enabled;
}

// Conv3d(64 -> 144, (1,3,3)), stride 1, padding (0,1,1), forward tap order (tap t = slab t at offset (t / 3 - 1, t % 3 - 1)).
bool cl16_sr_applies(const ClConv& g) {
  if (!sr_enabled() || g.ntaps != 9) return false;
  if (g.Cin_p != SR_CIN || g.Mrows != SR_COUT || g.Cout != SR_COUT || g.Cout_p != SR_COUTP) return false;
  if (g.Lt != g.Ti || g.Lh != g.Hi || g.Lw != g.Wi || g.To != g.Ti || g.Ho != g.Hi || g.Wo != g.Wi) return false;
  if (g.bmt != 1 || g.bmh != 1 || g.bmw != 1 || g.omt != 1 || g.omh != 1 || g.omw != 1 || g.oot || g.ooh || g.oow) return false;
  for (int t = 0; t < 9; ++t) {
    const int dt = (g.tap[t] & 15) - 8 + g.bot, dh = ((g.tap[t] >> 4) & 15) - 8 + g.boh, dw = ((g.tap[t] >> 8) & 15) - 8 + g.bow;
    if (dt != 0 || dh != t / 3 - 1 || dw != t % 3 - 1 || (g.tap[t] >> 12) != t) return false;
  }
  if ((long long)g.N * g.Ti * g.Hi * g.Wi * SR_COUTP * 2 >= 0xFFFFFFF0LL) return false;
  return true;
}

static int sr_tiles(const ClConv& g) { return g.N * g.Ti * ((g.Hi + SR_T - 1) / SR_T) * ((g.Wi + SR_T - 1) / SR_T); }
static int sr_grid(const ClConv& g) {
  static const int blocks = []() {
    const char* e = getenv("SELAVI_CL16_SR_BLOCKS");
    return e ? atoi(e) : 256;                         // one persistent workgroup per CU
  }();
  const int t = sr_tiles(g);
  return t < blocks ? t : blocks;
}
int cl16_sr_slots(const ClConv& g) { return sr_grid(g); }

template <int PRO, int EPI>
static int sr_launch_one(const ClConv& g, const void* x, const void* wl, void* y, const float* in_ss, float* stat_sum,
                         float* stat_sq, hipStream_t st) {
  static bool attr_set = false;
  if (!attr_set) {
    SLV_HIP(hipFuncSetAttribute((const void*)conv_cl16_sr_kernel<PRO, EPI>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                160 * 1024));
    attr_set = true;
  }
  const int th = (g.Hi + SR_T - 1) / SR_T, tw = (g.Wi + SR_T - 1) / SR_T;
  hipLaunchKernelGGL((conv_cl16_sr_kernel<PRO, EPI>), dim3(sr_grid(g)), dim3(256), SR_LDS, st, (const unsigned short*)x,
                     (const unsigned short*)wl, (unsigned short*)y, in_ss, stat_sum, stat_sq, g, sr_tiles(g), th, tw);
  return 0;
}

// returns 1 when the launch was taken, 0 when it does not apply, < 0 on error
int cl16_sr_try(const ClConv& g, const void* x, const void* wl, void* y, const float* in_ss, const float* scale_shift,
                const void* res, int relu, float* stat_sum, float* stat_sq, const ClBnr& bnr, hipStream_t st) {
  if (!cl16_sr_applies(g)) return 0;
  if (bnr.part) return fail(-2, "%s: the register-resident conv has no fused BatchNorm-backward sums", "slv_cl16_conv");
  if (scale_shift || res || relu) return 0;           // eval-mode epilogues: the tile kernel (no statistics involved)
  int rc = 0;
  const int pro = in_ss ? 1 : 0, epi = stat_sum ? 1 : 0;
  if (pro == 0 && epi == 0) rc = sr_launch_one<0, 0>(g, x, wl, y, in_ss, stat_sum, stat_sq, st);
  else if (pro == 1 && epi == 0) rc = sr_launch_one<1, 0>(g, x, wl, y, in_ss, stat_sum, stat_sq, st);
  else if (pro == 0 && epi == 1) rc = sr_launch_one<0, 1>(g, x, wl, y, in_ss, stat_sum, stat_sq, st);
  else rc = sr_launch_one<1, 1>(g, x, wl, y, in_ss, stat_sum, stat_sq, st);
  if (rc) return rc;
  rc = launch_check("slv_cl16_conv");
  return rc ? rc : 1;
}

}  // namespace slv
