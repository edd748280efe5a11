// 16-bit MFMA path: WEIGHT GRADIENT of the layer-1 spatial convolution (Conv3d 64 -> 144, (1,3,3), stride 1, padding
// (0,1,1); /root/reference/model.py:147-176 builds torchvision's r2plus1d_18, main.py:296-299 runs its backward) with the
// WHOLE dW resident in the accumulators of one workgroup.
//
// dW[co][ci][kh][kw] = sum over positions p of dY[p][co] * act(X)[p + (kh-1, kw-1)][ci] is a 144 x 576 matrix = 81 x 4
// MFMA tiles of 16 x 16: tiny next to the 12.8 M positions it is contracted over.  The patch kernel
// (csrc/wgrad_cl16_s3.hip) splits it into (Cout tile) x (kernel row) x (K slice) blocks, so dY is streamed three times and
// every 32-position step pays a barrier for 30 MFMAs.  Here the roles are turned around, as in the register-resident
// forward / backward-data kernels (conv_cl16_sr.hip, conv_cl16_sd.hip):
//   * a persistent workgroup (one per CU, 4 waves) owns ALL of dW for its share of the positions: wave w holds input
//     channels 16 w .. 16 w + 15 x 9 taps x 9 output-channel tiles = 81 accumulator tiles = 324 registers, 64 tiles in
//     the accumulator file ("+a" operands of an inline-asm MFMA), 17 in arch VGPRs;
//   * positions arrive as 8 x 8-pixel tiles: dY [64][160] by LDS-DMA (rows padded to 352 bytes = 32 (mod 64) for the
//     transpose reads), X as a 10 x 10 patch (zeros outside the image, BatchNorm + ReLU of the producing layer applied on
//     the way, 160-byte pixels) -- every byte of dY and X is read from memory once (X: + the halo, from L2);
//   * a K step = 32 positions = 4 tile rows: 9 dY fragments + 9 X fragments (one per tap: a constant LDS offset) by
//     ds_read_b64_tr_b16, 81 MFMAs; fragments of the next K step / tap are requested between the MFMAs of this one;
//   * three LDS buffers per operand: during tile k the data of tile k + 2 is requested (start of the tile) and
//     stored (end of the tile), one barrier per tile;
//   * at the end each workgroup writes its fp32 partial [co][tap][ci]; cl16_wgrad_reduce_kernel sums the partials in a
//     fixed order (deterministic).
// Per tile and wave: 162 MFMAs (2 600 cycles) against 33 KB from memory per CU: at 256 CUs that is the HBM roofline
// within 10 % of the MFMA one -- the kernel is bound by both.
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

#ifndef SLV_WA_ABL
#define SLV_WA_ABL 0               // timing ablations: 1 no MFMA, 2 no dY DMA, 3 no X loads, 4 neither, 5 no fragment reads
#endif
constexpr int WA_CIN = 64, WA_COUT = 144, WA_COUTP = 160;
constexpr int WA_XPB = 160;                    // bytes per patch pixel (128 + 32: 8 consecutive pixels -> 8 bank groups)
constexpr int WA_XBUF = 128 * WA_XPB;          // 100 live pixels; the 4th staging round of 224 idle threads lands behind them
constexpr int WA_YPB = 352;                    // bytes per dY pixel (320 + 32)
constexpr int WA_YBUF = 64 * WA_YPB;           // 22 528 = 22 DMA instructions of 1 KiB
constexpr int WA_NDMA = 22;
constexpr int WA_LDS = 3 * WA_XBUF + 3 * WA_YBUF;      // 129 024
constexpr int WA_NA = 64;                      // accumulator tiles in AGPRs (of 81)

__device__ __forceinline__ void wa_dma16(unsigned lds_addr, unsigned voff, __amdgpu_buffer_rsrc_t rsrc) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen" SLV_DMA_NT_STR " lds" ::"s"(lds_addr), "v"(voff), "s"(rsrc) : "m0", "memory");
}

struct WaTile {
  unsigned xbase, ybase;       // byte offsets of the tile's first pixel in X / dY (X: of pixel (-1,-1), wraps for border tiles)
  unsigned tm;                 // border bits of the tile (1 top, 2 bottom, 4 left, 8 right), 16 always, 32 = no such tile
  int live;
};

template <int PRO>
__global__ __launch_bounds__(256, 1) void cl16_wgrad_acc_kernel(const unsigned short* __restrict__ dy,
                                                                const unsigned short* __restrict__ x,
                                                                const float* __restrict__ in_ss, float* __restrict__ part,
                                                                int F, int H, int W, int TH, int TW, unsigned twmagic) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  typedef __attribute__((address_space(3))) void* lds_void;
  typedef __attribute__((address_space(3))) s16x4* lds_s16x4;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 15, fk = lane >> 4;
  unsigned char* const xlds = lds;
  unsigned char* const ylds = lds + 3 * WA_XBUF;
  const unsigned lds_base = (unsigned)(unsigned long)(lds_void)lds;
  const unsigned P = (unsigned)F * H * W;
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, (int)(P * (WA_CIN * 2u)), 0x00020000);
  const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc((void*)dy, 0, (int)(P * (WA_COUTP * 2u)), 0x00020000);
  const int TPF = TH * TW, ntiles = F * TPF, G = gridDim.x;
  const int nst = (int)blockIdx.x < ntiles ? (ntiles - 1 - (int)blockIdx.x) / G + 1 : 0;
  // tile sequence of this workgroup: id = blockIdx.x + k G, kept as (frame, tile in frame) and advanced without divisions
  const int stepf = G / TPF, stepr = G - stepf * TPF;
  int qf = (int)blockIdx.x / TPF, qr = (int)blockIdx.x - qf * TPF, qk = 0;          // the NEXT tile to be described
  auto next_tile = [&]() __attribute__((always_inline)) {
    WaTile t;
    const int ty = (int)(((unsigned)qr * twmagic) >> 16), tx = qr - ty * TW;
    t.live = qk < nst;
    const unsigned pix = ((unsigned)qf * H + ty * 8) * W + tx * 8;
    t.ybase = pix * (WA_COUTP * 2u);
    t.xbase = (pix - (unsigned)W - 1u) * (WA_CIN * 2u);
    t.tm = 16u | (ty == 0 ? 1u : 0u) | (ty == TH - 1 ? 2u : 0u) | (tx == 0 ? 4u : 0u) | (tx == TW - 1 ? 8u : 0u) |
           (t.live ? 0u : 32u);
    ++qk;
    qf += stepf;
    qr += stepr;
    if (qr >= TPF) {
      qr -= TPF;
      ++qf;
    }
    return t;
  };

  // ---- X staging: piece q = tid + 256 i of the patch (pixel q >> 3 = 10 py + px, 16-byte piece q & 7 = tid & 7)
  const int c8 = tid & 7;
  unsigned xoff[4], xflags = 0;
  int xdst[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int q = tid + 256 * i, pp = q >> 3, py = pp / 10, px = pp - py * 10;
    xoff[i] = (unsigned)((py * W + px) * (WA_CIN * 2) + c8 * 16);
    xdst[i] = pp * WA_XPB + c8 * 16;
    const unsigned fl = (py == 0 ? 1u : 0u) | (py == 9 ? 2u : 0u) | (px == 0 ? 4u : 0u) | (px == 9 ? 8u : 0u) |
                        (pp >= 100 ? 16u : 0u) | 32u;
    xflags |= fl << (8 * i);
  }
  float ps[8], ph[8];
  if constexpr (PRO == 1) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      ps[i] = in_ss[c8 * 8 + i];
      ph[i] = in_ss[WA_CIN + c8 * 8 + i];
    }
  }
  u32x4 xr[4];
  unsigned xok = 0;                            // bit i: piece i of the staged tile is inside the image
  auto x_load = [&](int i, const WaTile& t) __attribute__((always_inline)) {
    const bool ok = (((xflags >> (8 * i)) & 63u) & t.tm) == 0u;
    xok = (xok & ~(1u << i)) | ((unsigned)ok << i);
    if (SLV_WA_ABL == 3 || SLV_WA_ABL == 4) xr[i] = (u32x4){t.xbase, 1u, 2u, 3u};
    else {
      // inline asm: the compiler must not count these requests -- its own s_waitcnt for them (vmcnt(3..0), it does not see the
      // DMAs issued behind them) would drain the dY DMAs of the same tile at 60 % of the tile instead of at its end; the
      // hand-counted wait sits in front of the first staging item
      const unsigned voff = ok ? t.xbase + xoff[i] : 0xFFFFFFF0u;
      asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(xr[i]) : "v"(voff), "s"(rx) : "memory");
    }
  };
  u32x4 xo;
  // staging items of piece i: d = 0..3 one dword (two channels) each, d = 4 the store
  auto x_item = [&](int i, int d, int buf) __attribute__((always_inline)) {
    if constexpr (PRO == 1) {
      if (d < 4) {
        xo[d] = relu_bf2(pack_bf2(bn_affine(bf_lo(xr[i][d]), ps[2 * d], ph[2 * d]), bn_affine(bf_hi(xr[i][d]), ps[2 * d + 1], ph[2 * d + 1])));
      } else {
        const bool ok = (xok >> i) & 1u;
        const u32x4 v = ok ? xo : (u32x4){0u, 0u, 0u, 0u};
        *(u32x4*)(xlds + buf * WA_XBUF + xdst[i]) = v;
      }
    } else {
      if (d == 4) *(u32x4*)(xlds + buf * WA_XBUF + xdst[i]) = xr[i];       // (out-of-image requests returned zeros)
    }
  };

  // ---- dY by LDS-DMA: instruction jj = wave + 4 j covers the LDS pieces jj * 64 + lane (pixel = piece / 22)
  unsigned yo[6];
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    const int gp = (wave + 4 * j) * 64 + lane, px = gp / 22, col = gp - px * 22;
    yo[j] = col < 20 ? (unsigned)(((px >> 3) * W + (px & 7)) * (WA_COUTP * 2) + col * 16) : 0xFFFFFFF0u;
  }
  auto y_dma = [&](int j, int buf, const WaTile& t) __attribute__((always_inline)) {
    if (SLV_WA_ABL == 2 || SLV_WA_ABL == 4) return;
    if (wave + 4 * j < WA_NDMA) {
      const unsigned la = lds_base + 3u * WA_XBUF + (unsigned)buf * WA_YBUF + (unsigned)(wave + 4 * j) * 1024u;
      wa_dma16(__builtin_amdgcn_readfirstlane(la), (t.live && yo[j] != 0xFFFFFFF0u) ? t.ybase + yo[j] : 0xFFFFFFF0u, ry);
    }
  };

  // ---- fragments (ds_read_b64_tr_b16: lane (fr, fk) supplies the address of position 4 fk + (fr >> 2) (+16), 4 channels
  // at 4 (fr & 3); it receives channel fr at the positions 4 fk .. 4 fk + 3 (+16))
  const int p0 = 4 * fk + (fr >> 2);
  const int xlane = ((p0 >> 3) * 10 + (p0 & 7)) * WA_XPB + wave * 32 + 8 * (fr & 3);
  const int ylane = p0 * WA_YPB + 8 * (fr & 3);
  auto rd2 = [&](const unsigned char* lo_p, int hi_off) __attribute__((always_inline)) {
    if (SLV_WA_ABL == 5) return __builtin_bit_cast(bf16x8, (u32x4){(unsigned)(unsigned long)lo_p, 1u, 2u, (unsigned)hi_off});
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(lo_p));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(lo_p + hi_off));
    return tr_pair(lo, hi);
  };
  auto rd_x = [&](int buf, int ks, int t) __attribute__((always_inline)) {       // tap t = (kh, kw) = (t / 3, t % 3)
    return rd2(xlds + buf * WA_XBUF + xlane + (ks * 40 + (t / 3) * 10 + (t % 3)) * WA_XPB, 20 * WA_XPB);
  };
  auto rd_y = [&](int buf, int ks, int cot) __attribute__((always_inline)) {
    return rd2(ylds + buf * WA_YBUF + ylane + ks * 32 * WA_YPB + cot * 32, 16 * WA_YPB);
  };

  f32x4 accA[WA_NA], accV[81 - WA_NA];
#pragma unroll
  for (int m = 0; m < WA_NA; ++m) accA[m] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int m = 0; m < 81 - WA_NA; ++m) accV[m] = (f32x4){0.f, 0.f, 0.f, 0.f};
  bf16x8 dyf[2][9], xf[2];

  // One tile = 2 K steps x 9 taps x 9 output-channel tiles.  Slot = the place behind one MFMA.  Of a tap's 9 slots, slot 0
  // requests the X fragment of the next tap, slot 1 the dY fragment `tap` of the next K step; the other 7 x 18 = 126
  // slots carry the data movement of tile k + 2 (fs = free-slot index): 4 X requests, 6 DMA instructions, late in the
  // tile the 20 staging items.
#ifdef SLV_WA_TRACE   // s_memtime at 5 points of tiles 8..23, every wave of blocks 0..15 -> the head of `part` (timing only!)
#define WA_T(slot) if (tr_on && k >= 8 && k < 24) trace[(k - 8) * 5 + (slot)] = __builtin_amdgcn_s_memtime()
  const bool tr_on = lane == 0 && blockIdx.x < 16;
  unsigned long long* trace = (unsigned long long*)part + (size_t)((blockIdx.x * 4 + wave) * 16) * 5;
#else
#define WA_T(slot)
#endif
  auto tile_step = [&](int k, int bc, int bn, int bf, const WaTile& tf) __attribute__((always_inline)) {
    WA_T(0);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int cot = 0; cot < 9; ++cot) {
          const int m = t * 9 + cot, gt = ks * 9 + t;
          if (ks == 1 && m == 0) WA_T(1);
          if (SLV_WA_ABL == 1) {
            if (m < WA_NA) accA[m][0] += __builtin_bit_cast(float, __builtin_bit_cast(u32x4, dyf[ks][cot])[0] ^ __builtin_bit_cast(u32x4, xf[gt & 1])[0]);
            else accV[m - WA_NA][0] += __builtin_bit_cast(float, __builtin_bit_cast(u32x4, dyf[ks][cot])[0] ^ __builtin_bit_cast(u32x4, xf[gt & 1])[0]);
          } else if (m < WA_NA) {
            asm("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(accA[m]) : "v"(dyf[ks][cot]), "v"(xf[gt & 1]));
          } else {
            asm("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(accV[m - WA_NA]) : "v"(dyf[ks][cot]), "v"(xf[gt & 1]));
          }
          if (cot == 0) {
            if (t < 8) xf[(gt + 1) & 1] = rd_x(bc, ks, t + 1);
            else if (ks == 0) xf[(gt + 1) & 1] = rd_x(bc, 1, 0);
            else xf[(gt + 1) & 1] = rd_x(bn, 0, 0);
          } else if (cot == 1) {
            if (ks == 0) dyf[1][t] = rd_y(bc, 1, t);
            else dyf[0][t] = rd_y(bn, 0, t);
          } else {
            const int fs = ks * 63 + t * 7 + (cot - 2);
            if (fs < 4) x_load(fs, tf);
            else if (fs < 10) y_dma(fs - 4, bf, tf);
            else if (fs == 78) {                       // the four X requests have arrived; this wave's DMAs behind them may fly
              if (SLV_WA_ABL == 2 || SLV_WA_ABL == 4) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
              else if (wave < 2) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
              else asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
            } else if (fs >= 80 && fs < 120 && ((fs - 80) & 1) == 0) x_item((fs - 80) / 10, ((fs - 80) >> 1) % 5, bf);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
    WA_T(2);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    WA_T(3);
    __builtin_amdgcn_s_barrier();
    WA_T(4);
  };

  // ---- pipeline head: tiles 0 and 1 staged, then the first fragments
  {
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const WaTile t = next_tile();
#pragma unroll
      for (int i = 0; i < 4; ++i) x_load(i, t);
#pragma unroll
      for (int j = 0; j < 6; ++j) y_dma(j, b, t);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);          // (the asm orders memory operations only: pin the register uses behind it)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int d = 0; d < 5; ++d) x_item(i, d, b);
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int c = 0; c < 9; ++c) dyf[0][c] = rd_y(0, 0, c);
    xf[0] = rd_x(0, 0, 0);
  }
  int bc = 0, bn = 1, bf = 2;
  for (int k = 0; k < nst; ++k) {
    const WaTile tf = next_tile();
    tile_step(k, bc, bn, bf, tf);
    const int b0 = bc;
    bc = bn;
    bn = bf;
    bf = b0;
  }

  // ---- this workgroup's partial: part[wg][co][tap * 64 + ci]; C/D layout: column (ci) = lane & 15, rows (co) 4 fk + r
  asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");
  float* pw = part + (size_t)blockIdx.x * (WA_COUT * 9 * WA_CIN) + wave * 16 + fr;
#ifdef SLV_WA_TRACE
  if (accV[0][0] == 123.456f)
#endif
#pragma unroll
  for (int m = 0; m < 81; ++m) {
    const int t = m / 9, cot = m % 9;
    const f32x4 v = m < WA_NA ? accA[m < WA_NA ? m : 0] : accV[m < WA_NA ? 0 : m - WA_NA];
#pragma unroll
    for (int r = 0; r < 4; ++r) pw[(size_t)((cot * 16 + 4 * fk + r) * 9 + t) * WA_CIN] = v[r];
  }
}

static bool wa_enabled() {
  static const bool enabled = []() {
    const char* e = getenv("SELAVI_CL16_WGACC");
    return !(e && e[0] == '0');
  }();
  return enabled;
}

int wgrad_acc_blocks() {
  static const int blocks = []() {
    const char* e = getenv("SELAVI_CL16_WGACC_BLOCKS");
    const int b = e ? atoi(e) : 256;
    return b < 1 ? 1 : b;
  }();
  return blocks;
}

// Conv3d(64 -> 144, (1,3,3), stride 1, padding (0,1,1)) on whole 8 x 8 tiles, enough tiles to keep every workgroup busy
bool wgrad_acc_applies(int N, int T, int H, int W, int Cin_p, int Cin, int Cout_p, int Cout, int kt, int kh, int kw, int st,
                       int sh, int sw, int pt, int ph, int pw, int To, int Ho, int Wo) {
  if (!wa_enabled()) return false;
  if (kt != 1 || kh != 3 || kw != 3 || st != 1 || sh != 1 || sw != 1 || pt != 0 || ph != 1 || pw != 1) return false;
  if (To != T || Ho != H || Wo != W) return false;
  if (Cin_p != WA_CIN || Cin != WA_CIN || Cout_p != WA_COUTP || Cout != WA_COUT) return false;
  if ((H % 8) || (W % 8) || (H / 8) * (W / 8) > 1024 || W / 8 > 32) return false;
  const long long ntiles = (long long)N * T * (H / 8) * (W / 8);
  if (ntiles < 8LL * wgrad_acc_blocks()) return false;
  if ((long long)N * T * H * W * (WA_COUTP * 2) >= 0xFFFFFFF0LL) return false;
  return true;
}

size_t wgrad_acc_ws_bytes() { return (size_t)wgrad_acc_blocks() * WA_COUT * 9 * WA_CIN * sizeof(float); }

int wgrad_acc_launch(int N, int T, int H, int W, const void* dy, const void* x, const float* in_ss, float* part, hipStream_t st) {
  static bool attr_set = false;
  if (!attr_set) {
    SLV_HIP(hipFuncSetAttribute((const void*)cl16_wgrad_acc_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    SLV_HIP(hipFuncSetAttribute((const void*)cl16_wgrad_acc_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_set = true;
  }
  const int TH = H / 8, TW = W / 8;
  const unsigned twmagic = 65536u / (unsigned)TW + 1u;
  const dim3 grid(wgrad_acc_blocks());
  if (in_ss)
    hipLaunchKernelGGL((cl16_wgrad_acc_kernel<1>), grid, dim3(256), WA_LDS, st, (const unsigned short*)dy,
                       (const unsigned short*)x, in_ss, part, N * T, H, W, TH, TW, twmagic);
  else
    hipLaunchKernelGGL((cl16_wgrad_acc_kernel<0>), grid, dim3(256), WA_LDS, st, (const unsigned short*)dy,
                       (const unsigned short*)x, in_ss, part, N * T, H, W, TH, TW, twmagic);
  return launch_check("slv_cl16_wgrad");
}

}  // namespace slv
