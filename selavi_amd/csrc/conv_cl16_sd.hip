// 16-bit MFMA path: BACKWARD DATA of the layer-1 spatial convolution -- the (1,3,3) conv 144 (stored 160) -> 64 on flipped
// taps -- the longest kernel on the critical path of the step (K = 9 x 160: two thirds more work than the forward), with
// the WEIGHTS RESIDENT IN REGISTERS and the input patches fed by LDS-DMA.
//
// Same idea as csrc/conv_cl16_sr.hip (the weight tensor is tiny, the activations are huge: keep the weights in the
// register file for the kernel's lifetime and stream 8 x 8-pixel tiles through), other numbers:
//   * 9 taps x 5 chunks x 4 output tiles = 180 A fragments = 720 registers do not fit one wave: a wave owns HALF the
//     output channels (2 tiles of 16) = 90 fragments = 360 registers -- 64 of them in the accumulator file ("a"
//     operands), 26 in the arch VGPRs;
//   * a workgroup (one per CU) is 4 MFMA waves = 2 pairs; a pair works on its own tile (its two waves = the two channel
//     halves, reading the same patch), so 128 pixels per workgroup step and all four matrix cores busy;
//   * no wave has registers to stage a 32 KB patch: the patch of the NEXT tile goes memory -> LDS by LDS-DMA
//     (buffer_load ... lds, issued between the MFMAs by the pair's two waves, 16 instructions each).  The DMA is inline
//     asm: left to the compiler, every LDS read it can see is ordered behind a DMA "that may alias" with vmcnt(0).  The
//     tile ends with s_waitcnt vmcnt(0) + barrier (the DMAs were issued in the tile's first third);
//   * LDS patch: 10 x 10 pixels x 320-byte rows, compact; the 16-byte slot of (chunk c, k-group q) of patch column px is
//     4 c + (q ^ (px & 3)) -- found by enumeration to be conflict-free for ds_read_b128 fragments of 2 tile rows x 8
//     columns at every tap shift.  A DMA writes LDS linearly, so the XOR sits on the SOURCE address; the reader keeps
//     three lane bases (one per tap column) and every tap row / chunk / fragment is an immediate offset;
//   * the epilogue of tile n - 1 (bf16 through a wave-private LDS stage, 64-byte half rows to memory) rides between
//     the MFMAs of tile n (two accumulator sets).
// An in-order wave cannot issue past a waiting MFMA, so everything that is not an MFMA is cut into items of <= 3-4
// instructions and placed one per MFMA slot.
#include "cl16.hpp"
#include "../../include/selavi_hip.h"

#ifndef SLV_DMA_NT
#define SLV_DMA_NT 0      // 1: the LDS-DMA requests of the streamed operand carry the non-temporal hint (A/B: r05 notes)
#endif
#if SLV_DMA_NT
#define SLV_DMA_NT_STR " nt"
#else
#define SLV_DMA_NT_STR ""
#endif

namespace slv {

constexpr int SD_T = 8, SD_PW = 10;
constexpr int SD_CINP = 160, SD_KC = 5, SD_COUT = 64;
constexpr int SD_ROWB = SD_CINP * 2;                  // 320
constexpr int SD_PATCH = 32 * 1024;                   // 100 pixels x 320 B = 32 000, rounded up to 32 DMA instructions
constexpr int SD_OROW = 64 + 16;                      // bytes per pixel row of a wave's output stage (32 channels)
constexpr int SD_OST = SD_T * SD_T * SD_OROW;         // 5 120
constexpr int SD_GTAB = 2 * 16 * 64 * 4;              // per channel half: [16 DMA instructions][64 lanes] source offsets
constexpr int SD_LDS = 4 * SD_PATCH + 4 * SD_OST + SD_GTAB;     // 159 744
constexpr int SD_NA = 64;                             // A fragments held in AGPRs (the other 26: arch VGPRs)
constexpr int SD_DMA = 16;                            // DMA instructions per wave and patch

__device__ __forceinline__ void sd_mfma_a(f32x4& acc, const bf16x8& a, const bf16x8& b) {
  asm("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc) : "a"(a), "v"(b));
}
__device__ __forceinline__ void sd_mfma_v(f32x4& acc, const bf16x8& a, const bf16x8& b) {
  asm("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
}
__device__ __forceinline__ void sd_mfma0_a(f32x4& acc, const bf16x8& a, const bf16x8& b) {
  asm("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=v"(acc) : "a"(a), "v"(b));
}
// one LDS-DMA instruction: 64 lanes x 16 bytes from memory (per-lane byte offset voff into the buffer rsrc) to the
// wave-uniform LDS address lds_addr + 16 lane.  Invisible to the compiler's wait-count bookkeeping (on purpose).
__device__ __forceinline__ void sd_dma16(unsigned lds_addr, unsigned voff, __amdgpu_buffer_rsrc_t rsrc) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen" SLV_DMA_NT_STR " lds" ::"s"(lds_addr), "v"(voff), "s"(rsrc) : "m0", "memory");
}

struct SdTile {
  int y0, x0;
  unsigned fpos;
  int live;
};

// RES 1: y = acc + res (the gradient that arrives over the residual connection; res may be y itself), one rounding
template <int RES>
__global__ __launch_bounds__(256, 1) void conv_cl16_sd_kernel(const unsigned short* __restrict__ x,
                                                             const unsigned short* __restrict__ wl,
                                                             unsigned short* y, const unsigned short* res, ClConv g,
                                                             int ntiles, int th, int tw) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int pr = wave >> 1, hf = wave & 1;            // pair (its own tile), channel half
  const int fr = lane & 15, fk = lane >> 4;
  const int H = g.Hi, W = g.Wi;
  unsigned char* const pbuf = lds + pr * (2 * SD_PATCH);                 // the pair's two patch buffers
  unsigned char* const ost = lds + 4 * SD_PATCH + wave * SD_OST;        // this wave's output stage
  typedef __attribute__((address_space(3))) void* lds_void;
  const unsigned lds_base = (unsigned)(unsigned long)(lds_void)lds;
  const unsigned Ptot = (unsigned)g.N * g.Ti * H * W;
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, (int)(Ptot * (unsigned)SD_ROWB), 0x00020000);
  const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc((void*)y, 0, (int)(Ptot * (SD_COUT * 2u)), 0x00020000);
  // steps: the workgroup takes tile pairs (2 k, 2 k + 1) of its sequence; this wave's pair takes one of the two
  const int npairs = (ntiles + 1) >> 1;
  const int nst = blockIdx.x < (unsigned)npairs ? (npairs - 1 - (int)blockIdx.x) / (int)gridDim.x + 1 : 0;
  // (tile id -> frame / row / column by multiply-high with the reciprocals, exact for ids < 2^32 / divisor: two tile decodes per
  //  step as integer divisions were ~140 instructions in front of the step's MFMAs)
  const unsigned per_u = (unsigned)(th * tw), mper = cl_recip(per_u), mtw = cl_recip((unsigned)tw);
  auto tile_of = [&](int k) __attribute__((always_inline)) {
    SdTile t;
    const int id = (blockIdx.x + k * gridDim.x) * 2 + pr;
    const int per = (int)per_u, f = (int)cl_div((unsigned)id, mper), rem = id - f * per, ty = (int)cl_div((unsigned)rem, mtw);
    t.y0 = ty * SD_T;
    t.x0 = (rem - ty * tw) * SD_T;
    t.fpos = (unsigned)f * H * W;
    t.live = k < nst && id < ntiles;
    return t;
  };

  // ---- the weights: tap t = slab t at offset -(t / 3 - 1, t % 3 - 1); fragment index fi = (t * 5 + c) * 2 + i
  bf16x8 Aa[SD_NA], Av[90 - SD_NA];
  auto a_frag = [&](int fi) __attribute__((always_inline)) {
    const int t = fi / 10, c = (fi / 2) % 5, i = fi & 1;
    return *(const bf16x8*)(wl + ((size_t)((t * SD_KC + c) * SD_COUT + (hf * 2 + i) * 16 + fr) * 32 + fk * 8));
  };
#pragma unroll
  for (int fi = 0; fi < SD_NA; ++fi) Aa[fi] = a_frag(fi);
#pragma unroll
  for (int fi = SD_NA; fi < 90; ++fi) Av[fi - SD_NA] = a_frag(fi);
  // ---- fragment reads: patch pixel of (fragment nn, lane, tap) = (2 nn + (fr >> 3) + dy, (fr & 7) + dx), dy, dx in 0..2;
  // one lane base per dx (the XOR key is the patch column & 3)
  int bbase[3];
#pragma unroll
  for (int o = 0; o < 3; ++o) {
    const int px = (fr & 7) + o;
    bbase[o] = ((fr >> 3) * SD_PW + px) * SD_ROWB + ((fk ^ (px & 3)) << 4);
  }
  // ---- this wave's DMA pieces: instruction j covers LDS pieces (hf * 16 + j) * 64 + lane of the patch
  // (their source offsets relative to the patch origin live in an LDS table -- 16 registers the wave does not have --
  //  written once; the border masks are 16-bit sets)
  unsigned* const gtab = (unsigned*)(lds + 4 * SD_PATCH + 4 * SD_OST) + hf * (SD_DMA * 64);
  unsigned mtop = 0, mbot = 0, mleft = 0, mright = 0, mlive = 0;
#pragma unroll
  for (int j = 0; j < SD_DMA; ++j) {
    const int gp = (hf * SD_DMA + j) * 64 + lane;     // LDS piece index
    const int idx = gp / 20, sp = gp - idx * 20;      // patch pixel, slot within its row
    const int py = idx / SD_PW, px = idx - py * SD_PW;
    const int c = sp >> 2, q = (sp & 3) ^ (px & 3);   // the memory piece that belongs into this slot
    const bool live = idx < SD_PW * SD_PW;
    if (pr == 0) gtab[j * 64 + lane] = (unsigned)((py * W + px) * SD_ROWB + c * 64 + q * 16);
    mlive |= (unsigned)live << j;
    mtop |= (unsigned)(py == 0) << j;
    mbot |= (unsigned)(py == SD_PW - 1) << j;
    mleft |= (unsigned)(px == 0) << j;
    mright |= (unsigned)(px == SD_PW - 1) << j;
  }
  __syncthreads();
  unsigned dvalid = 0, dbase = 0;
  auto dma_begin = [&](const SdTile& t) __attribute__((always_inline)) {
    dvalid = t.live ? mlive & ~((t.y0 == 0 ? mtop : 0u) | (t.y0 + SD_T == H ? mbot : 0u) | (t.x0 == 0 ? mleft : 0u) |
                                (t.x0 + SD_T == W ? mright : 0u))
                    : 0u;
    dbase = (t.fpos + (unsigned)((t.y0 - 1) * W + (t.x0 - 1))) * (unsigned)SD_ROWB;      // (wraps; valid pieces only)
  };
  auto dma_item = [&](int j, int buf) __attribute__((always_inline)) {
    const unsigned la = lds_base + (unsigned)(pr * 2 + buf) * SD_PATCH + (unsigned)(hf * SD_DMA + j) * 1024u;
    sd_dma16(__builtin_amdgcn_readfirstlane(la), ((dvalid >> j) & 1) ? dbase + gtab[j * 64 + lane] : 0xFFFFFFF0u, rx);
  };
  // ---- output: 4 pieces (this wave's 32 channels) per pixel, 16 pixels per store
  const int spiece = lane & 3, spx = lane >> 2;
  const int obase = fr * SD_OROW + fk * 8;
  unsigned soff[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int p = k * 16 + spx;
    soff[k] = (unsigned)(((p >> 3) * W + (p & 7)) * (SD_COUT * 2) + hf * 64 + spiece * 16);
  }
  constexpr int SD_ITEMS = 16;
  u32x4 carry = {0u, 0u, 0u, 0u};
  // the addend in accumulator layout: pixel nn * 16 + fr, channels (hf * 2 + i) * 16 + 4 fk .. + 3 (8 bytes)
  const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc((void*)(RES ? res : y), 0, (int)(Ptot * (SD_COUT * 2u)), 0x00020000);
  unsigned aoff[4];
#pragma unroll
  for (int nn = 0; nn < 4; ++nn) {
    const int p = nn * 16 + fr;
    aoff[nn] = (unsigned)(((p >> 3) * W + (p & 7)) * (SD_COUT * 2) + hf * 64 + fk * 8);
  }
  uint2 radd[8];
  // RES: the addend of accumulator tile `it` (8 requests per tile), as INLINE ASM: the compiler must not count them -- its
  // own s_waitcnt for a visible load (vmcnt(7..0): it does not see the patch DMAs) drained the 16 DMAs of the same tile a
  // quarter into the tile instead of at its end (sd<1> 2.39 ms against sd<0> 2.05 ms in the cfg5 step).  They are issued
  // in FRONT of the DMAs; a hand-counted s_waitcnt vmcnt(16) (this wave's DMA instructions) precedes the first use.
  auto res_load = [&](int it, const SdTile& tl) __attribute__((always_inline)) {
    const unsigned tb = (tl.fpos + (unsigned)(tl.y0 * W + tl.x0)) * (SD_COUT * 2u);
    const unsigned voff = tl.live ? tb + aoff[it & 3] + (unsigned)((it >> 2) * 32) : 0xFFFFFFF0u;
    asm volatile("buffer_load_dwordx2 %0, %1, %2, 0 offen" : "=v"(radd[it]) : "v"(voff), "s"(rr) : "memory");
  };
  // items of the previous tile's epilogue, in order: one accumulator tile each (+ addend) -> bf16 -> stage; then store k =
  // read its pieces, issue it
  auto drain_item = [&](int it0, f32x4 (&pv)[2][4], const SdTile& tl) __attribute__((always_inline)) {
    int it = it0;
    if (it < 8) {
      const int i = it >> 2, nn = it & 3;
      float v[4] = {pv[i][nn][0], pv[i][nn][1], pv[i][nn][2], pv[i][nn][3]};
      if constexpr (RES == 1) {
        v[0] += bf_lo(radd[it].x);
        v[1] += bf_hi(radd[it].x);
        v[2] += bf_lo(radd[it].y);
        v[3] += bf_hi(radd[it].y);
      }
      const unsigned lo = pack_bf2(v[0], v[1]), hi = pack_bf2(v[2], v[3]);
      *(uint2*)(ost + obase + nn * 16 * SD_OROW + i * 32) = make_uint2(lo, hi);
    } else if (it < 16) {
      const int k = (it - 8) >> 1;
      if (((it - 8) & 1) == 0) {
        carry = *(const u32x4*)(ost + (k * 16 + spx) * SD_OROW + spiece * 16);
      } else {
        const unsigned tb = (tl.fpos + (unsigned)(tl.y0 * W + tl.x0)) * (SD_COUT * 2u);
        __builtin_amdgcn_raw_buffer_store_b128(carry, ry, tl.live ? tb + soff[k] : 0xFFFFFFF0u, 0, CL_NT);
      }
    }
  };
  // One tile: 45 MFMA groups (tap, chunk) of 8 MFMAs (2 channel tiles x 4 fragments).  Slot = the place behind one MFMA:
  // even slots of a group request the fragments of the NEXT group, odd slots carry one item each -- first the DMA of the
  // next patch (groups 0-3), then the previous tile's epilogue (two accumulator sets; groups 5-8).
  auto tile_step = [&](int n, f32x4 (&acc)[2][4], f32x4 (&pv)[2][4], bool drain) __attribute__((always_inline)) {
    const unsigned char* src = pbuf + (n & 1) * SD_PATCH;
    const SdTile tprev = tile_of(n > 0 ? n - 1 : 0);
    dma_begin(tile_of(n + 1));
    bf16x8 b[2][4];
    auto read_b = [&](int gi, int nn, bf16x8* to) __attribute__((always_inline)) {
      const int t = gi / SD_KC, c = gi - t * SD_KC;
      const int dy = 2 - t / 3, dx = 2 - t % 3;        // flipped taps: slab t at offset -(t / 3 - 1, t % 3 - 1)
      to[nn] = *(const bf16x8*)(src + bbase[dx] + ((2 * nn + dy) * SD_PW) * SD_ROWB + c * 64);
    };
#pragma unroll
    for (int nn = 0; nn < 4; ++nn) read_b(0, nn, b[0]);
#pragma unroll
    for (int gi = 0; gi < 45; ++gi) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int nn = 0; nn < 4; ++nn) {
          const int fi = gi * 2 + i;
          if (gi == 0) sd_mfma0_a(acc[i][nn], Aa[fi], b[0][nn]);
          else if (fi < SD_NA) sd_mfma_a(acc[i][nn], Aa[fi], b[gi & 1][nn]);
          else sd_mfma_v(acc[i][nn], Av[fi - SD_NA], b[gi & 1][nn]);
          const int ph = i * 4 + nn;                   // slot within the group
          if ((ph & 1) == 0) {
            if (gi + 1 < 45) read_b(gi + 1, ph >> 1, b[(gi + 1) & 1]);
          } else {
            const int fs = gi * 4 + (ph >> 1);         // free-slot index within the tile
            if constexpr (RES == 1) {
              if (fs < 8) {
                if (drain) res_load(fs, tprev);
              } else if (fs < 8 + SD_DMA) dma_item(fs - 8, (n + 1) & 1);
              else if (fs == 26) {
                if (drain) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
              } else if (fs >= 28 && fs < 28 + 2 * SD_ITEMS && ((fs - 28) & 1) == 0 && drain) drain_item((fs - 28) >> 1, pv, tprev);
            } else {
              if (fs < SD_DMA) dma_item(fs, (n + 1) & 1);
              else if (fs >= 20 && fs < 20 + 2 * SD_ITEMS && ((fs - 20) & 1) == 0 && drain) drain_item((fs - 20) >> 1, pv, tprev);
            }
          }
          __builtin_amdgcn_sched_barrier(0);
        }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");      // the next patch has landed (and this wave's stores)
    __builtin_amdgcn_s_barrier();
  };
  // ---- pipeline head: patch 0 by DMA, wait, barrier
  dma_begin(tile_of(0));
#pragma unroll
  for (int j = 0; j < SD_DMA; ++j) dma_item(j, 0);
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  f32x4 accA[2][4], accB[2][4];
  for (int n = 0; n < nst; n += 2) {
    tile_step(n, accA, accB, n > 0);
    if (n + 1 < nst) tile_step(n + 1, accB, accA, true);
  }
  if (nst > 0) {                                      // the last tile's epilogue
    mfma_settle_nops();
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int nn = 0; nn < 4; ++nn) mfma_pin(accA[i][nn]), mfma_pin(accB[i][nn]);
    const SdTile tl = tile_of(nst - 1);
    if constexpr (RES == 1) {
#pragma unroll
      for (int it = 0; it < 8; ++it) res_load(it, tl);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);          // (the asm orders memory operations only: pin the register uses behind it)
    }
    if ((nst - 1) & 1) {
#pragma unroll
      for (int it = 0; it < SD_ITEMS; ++it) drain_item(it, accB, tl);
    } else {
#pragma unroll
      for (int it = 0; it < SD_ITEMS; ++it) drain_item(it, accA, tl);
    }
  }
}

static bool sd_enabled() {
  static const bool enabled = []() {
    const char* e = getenv("SELAVI_CL16_SD");
    return !(e && e[0] == '0');
  }();
  return enabled;
}

// backward data of Conv3d(64 -> 144, (1,3,3), stride 1, padding (0,1,1)): 160 stored channels -> 64, taps in descending
// order (tap t = slab t at offset -(t / 3 - 1, t % 3 - 1)), whole 8 x 8 tiles
bool cl16_sd_applies(const ClConv& g) {
  if (!sd_enabled() || g.ntaps != 9) return false;
  if (g.Cin_p != SD_CINP || g.Mrows != SD_COUT || g.Cout != SD_COUT || g.Cout_p != SD_COUT) return false;
  if (g.Lt != g.Ti || g.Lh != g.Hi || g.Lw != g.Wi || g.To != g.Ti || g.Ho != g.Hi || g.Wo != g.Wi) return false;
  if (g.bmt != 1 || g.bmh != 1 || g.bmw != 1 || g.omt != 1 || g.omh != 1 || g.omw != 1 || g.oot || g.ooh || g.oow) return false;
  if ((g.Hi % SD_T) || (g.Wi % SD_T)) return false;
  for (int t = 0; t < 9; ++t) {
    const int dt = (g.tap[t] & 15) - 8 + g.bot, dh = ((g.tap[t] >> 4) & 15) - 8 + g.boh, dw = ((g.tap[t] >> 8) & 15) - 8 + g.bow;
    if (dt != 0 || dh != -(t / 3 - 1) || dw != -(t % 3 - 1) || (g.tap[t] >> 12) != t) return false;
  }
  if ((long long)g.N * g.Ti * g.Hi * g.Wi * SD_ROWB >= 0xFFFFFFF0LL) return false;
  {                                                   // the tile decode by multiply-high is exact for ids < 2^32 / (tiles per frame)
    const long long per = (long long)(g.Hi / SD_T) * (g.Wi / SD_T);
    if (per < 1 || ((long long)g.N * g.Ti * per + 4096) * per >= 0xFFFFFFFFLL) return false;
  }
  return true;
}

// returns 1 when the launch was taken, 0 when it does not apply (prologue / statistics / affine / fused sums: the tile
// kernel), < 0 on error
int cl16_sd_try(const ClConv& g, const void* x, const void* wl, void* y, const float* in_ss, const float* scale_shift,
                const void* res, int relu, float* stat_sum, float* stat_sq, const ClBnr& bnr, hipStream_t st) {
  if (!cl16_sd_applies(g)) return 0;
  if (in_ss || scale_shift || relu || stat_sum || bnr.part) return 0;
  static bool attr_set = false;
  if (!attr_set) {
    SLV_HIP(hipFuncSetAttribute((const void*)conv_cl16_sd_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    SLV_HIP(hipFuncSetAttribute((const void*)conv_cl16_sd_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_set = true;
  }
  const int th = g.Hi / SD_T, tw = g.Wi / SD_T, ntiles = g.N * g.Ti * th * tw;
  const int npairs = (ntiles + 1) / 2;
  static const int blocks = []() {
    const char* e = getenv("SELAVI_CL16_SD_BLOCKS");
    return e ? atoi(e) : 256;
  }();
  const dim3 grid(npairs < blocks ? npairs : blocks);
  if (res)
    hipLaunchKernelGGL((conv_cl16_sd_kernel<1>), grid, dim3(256), SD_LDS, st, (const unsigned short*)x,
                       (const unsigned short*)wl, (unsigned short*)y, (const unsigned short*)res, g, ntiles, th, tw);
  else
    hipLaunchKernelGGL((conv_cl16_sd_kernel<0>), grid, dim3(256), SD_LDS, st, (const unsigned short*)x,
                       (const unsigned short*)wl, (unsigned short*)y, (const unsigned short*)nullptr, g, ntiles, th, tw);
  const int rc = launch_check("slv_cl16_conv");
  return rc ? rc : 1;
}

}  // namespace slv
