// 16-bit MFMA path: the two products G2 = sum_p dY (m y), G1 = sum_p dY m of the layer-1 TEMPORAL convolution (Conv3d
// 144 -> 64, (3,1,1), stride 1; /root/reference/model.py:147-176, backward: main.py:296-299) with BOTH in the accumulators of
// one persistent workgroup -- the kernel that makes csrc/wgrad_cl16_t2.hip pay: dW = s G2 + h G1 is the weight gradient,
// sum W G1 / sum W G2 are the BatchNorm-backward sums of the spatial conv in front (no reduce pass over g and y).
//
// The column-order kernel (csrc/wgrad_cl16_t.hip) is bound by its per-step load -> store -> barrier chain (32 positions, 30
// MFMAs per wave and step); forming two products there doubled its time.  Here, as in csrc/wgrad_cl16_acc.hip:
//   * one workgroup per CU, 4 waves = (kind k) x (output-channel half h); a wave holds 2 co-tiles x 3 taps x 9 ci-tiles = 54
//     accumulator tiles = 216 registers, all in the accumulator file ("+a" operands of an inline-asm MFMA);
//   * a workgroup walks 32-pixel columns frame by frame; after the T frames of a column comes one VIRTUAL frame of zeros (it
//     is the "frame -1" of the next column and the "frame T" of this one: no special cases at clip borders);
//   * the X fragments of the frames t-1, t, t+1 live in a REGISTER RING (27 fragments): a step reads only the 9 fragments of
//     the new frame t+1 and the 2 dY fragments of frame t (22 transpose reads for 54 MFMAs); the taps -1 and 0 start on
//     fragments that are already there while the new ones arrive;
//   * a step is only 54 MFMAs (~0.4 us), far less than a trip to memory: y of step s+5 is REQUESTED during step s (inline-asm
//     loads into one of three register sets: the compiler neither waits for them nor moves them), STAGED during step s+3
//     behind a hand-counted s_waitcnt vmcnt(2 x the wave's requests per step) as TWO tiles, (m y) and m as bf16
//     (m = [y s + h > 0]; both exact), and its fragments are read during step s+4; dY of step s+5 by inline-asm LDS-DMA
//     into a ring of six tiles.  X tiles double-buffered, one barrier per step.
// Partials: part[kind][workgroup][64][3 * 160] -- the layout cl16_wgrad_t2_sum_kernel / _finish_kernel read.
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

#ifndef SLV_TA_ABL
#define SLV_TA_ABL 0               // timing ablations: 1 no mask math, 2 no global requests, 3 no MFMA, 4 no fragment reads
#endif
constexpr int TA_CINP = 160, TA_NCI = 9, TA_COUT = 64;
constexpr int TA_XPB = TA_CINP * 2 + 32;            // 352 bytes per pixel of an X tile (32 mod 64 for the transpose reads)
constexpr int TA_XTILE = 32 * TA_XPB;               // 11 264
constexpr int TA_XBUF = 2 * TA_XTILE;               // both kinds of one frame
constexpr int TA_YPB = TA_COUT * 2 + 32;            // 160 bytes per pixel of a dY tile
constexpr int TA_YBUF = 5 * 1024;                   // 32 x 160 = 5 120 = 5 DMA instructions
constexpr int TA_NYB = 6;                           // dY tiles in the ring
constexpr int TA_LDS = 2 * TA_XBUF + TA_NYB * TA_YBUF;   // 75 776

__device__ __forceinline__ void ta_dma16(unsigned lds_addr, unsigned voff, __amdgpu_buffer_rsrc_t rsrc) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen" SLV_DMA_NT_STR " lds" ::"s"(lds_addr), "v"(voff), "s"(rsrc) : "m0", "memory");
}

struct TaStep {
  unsigned row0;        // first tensor row (position) of the step's 32-pixel frame tile
  int nr;               // rows that exist: 0 for the virtual frame and beyond the sequence
};

__global__ __launch_bounds__(256, 1) void cl16_wgrad_tacc_kernel(const unsigned short* __restrict__ dy,
                                                                 const unsigned short* __restrict__ x,
                                                                 const float* __restrict__ in_ss, float* __restrict__ part,
                                                                 size_t kind_stride, int N, int T, int HW, int PB, int Cin) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  typedef __attribute__((address_space(3))) void* lds_void;
  typedef __attribute__((address_space(3))) s16x4* lds_s16x4;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int kind = wave & 1, hf = wave >> 1;
  const int fr = lane & 15, fk = lane >> 4;
  unsigned char* const xlds = lds;
  unsigned char* const ylds = lds + 2 * TA_XBUF;
  const unsigned lds_base = (unsigned)(unsigned long)(lds_void)lds;
  const unsigned P = (unsigned)N * T * HW;
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, (int)(P * (TA_CINP * 2u)), 0x00020000);
  const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc((void*)dy, 0, (int)(P * (TA_COUT * 2u)), 0x00020000);
  // this workgroup's columns: c = blockIdx.x + j gridDim.x; its step sequence: (j, tv), tv = 0 .. T (tv == T: virtual)
  const int ncol = N * PB, G = gridDim.x, T1 = T + 1;
  const int mycols = (int)blockIdx.x < ncol ? (ncol - 1 - (int)blockIdx.x) / G + 1 : 0;
  const int nsteps = mycols * T1;
  int qj = 0, qt = 0;                                   // the NEXT step to be described
  unsigned qrow = 0;                                    // first row of its column's frame 0
  int qnr = 0;                                          // rows of its column (0: no such column)
  auto column = [&]() __attribute__((always_inline)) { // (one integer division per column, not per step)
    const int col = (int)blockIdx.x + qj * G;
    const int n = col / PB, pb = col - n * PB, px0 = pb * 32;
    qrow = (unsigned)n * T * (unsigned)HW + (unsigned)px0;
    qnr = qj < mycols ? min(32, HW - px0) : 0;
  };
  column();
  auto next_step = [&]() __attribute__((always_inline)) {
    TaStep s;
    s.row0 = qrow + (unsigned)qt * (unsigned)HW;
    s.nr = qt < T ? qnr : 0;
    if (++qt == T1) {
      qt = 0;
      ++qj;
      column();
    }
    return s;
  };

  // ---- X staging: threads 0..239 take piece pc = tid % 20 (fixed: its 8 channels' scale / shift stay in registers) of the
  // pixels tid / 20 + 12 i, i = 0..2
  const int spc = tid % 20, spx = tid / 20;
  float ps[8], ph[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int c = spc * 8 + i;
    ps[i] = (tid < 240 && c < Cin) ? in_ss[c] : 0.f;
    ph[i] = (tid < 240 && c < Cin) ? in_ss[Cin + c] : 0.f;
  }
  // (a use in front of the loop: otherwise the compiler's own s_waitcnt vmcnt(0) for these ONE-TIME loads lands in front of
  //  their first use INSIDE the loop body, every iteration -- and drains the five steps of requests the pipeline keeps in flight)
#pragma unroll
  for (int i = 0; i < 8; ++i) asm volatile("" ::"v"(ps[i]), "v"(ph[i]));
  u32x4 xr[3][3];                                       // [register set][piece]: written by in-flight loads, hands off!
  unsigned xok = 0;                                     // bit set * 3 + i: that piece is inside the tensor
  auto x_load = [&](int S, int i, const TaStep& s) __attribute__((always_inline)) {      // S, i: compile time after inlining
    const int px = spx + 12 * i;
    const bool ok = tid < 240 && px < s.nr;
    xok = (xok & ~(1u << (S * 3 + i))) | ((unsigned)ok << (S * 3 + i));
    const unsigned voff = ok ? (s.row0 + (unsigned)px) * (TA_CINP * 2u) + (unsigned)spc * 16u : 0xFFFFFFF0u;
    if (SLV_TA_ABL == 2) xr[S][i] = (u32x4){voff, 1u, 2u, 3u};
    else asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(xr[S][i]) : "v"(voff), "s"(rx) : "memory");
  };
  // items of piece i: d = 0..3 the mask of one dword (two channels), d = 4 the two stores
  u32x4 xmy, xm1;
  auto x_item = [&](int S, int i, int d, int buf) __attribute__((always_inline)) {
    if (d < 4) {
      const bool live = (xok >> (S * 3 + i)) & 1u;
      const unsigned v = xr[S][i][d];
      // (no short-circuit "live && ...": hipcc turns it into a divergent branch per half; a dead piece's one is zeroed)
      const bool lo = SLV_TA_ABL == 1 ? true : bn_affine(bf_lo(v), ps[2 * d], ph[2 * d]) > 0.f;
      const bool hi = SLV_TA_ABL == 1 ? true : bn_affine(bf_hi(v), ps[2 * d + 1], ph[2 * d + 1]) > 0.f;
      const unsigned keep = (lo ? 0x0000FFFFu : 0u) | (hi ? 0xFFFF0000u : 0u);
      xmy[d] = v & keep;                                 // (a request outside the tensor returned zeros)
      xm1[d] = (live ? 0x3F803F80u : 0u) & keep;
    } else {
      const int px = spx + 12 * i;
      if (tid < 240 && px < 32) {
        unsigned char* dst = xlds + buf * TA_XBUF + px * TA_XPB + spc * 16;
        *(u32x4*)dst = xmy;
        *(u32x4*)(dst + TA_XTILE) = xm1;
      }
    }
  };

  // ---- dY by LDS-DMA: instruction jj covers LDS pieces jj * 64 + lane (pixel = piece / 10, 8 data pieces + 2 padding);
  // wave w issues jj = w and, wave 0, jj = 4
  unsigned yo[2];
  int ypx[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int gp = (wave + 4 * j) * 64 + lane, px = gp / 10, col = gp - px * 10;
    ypx[j] = px;
    yo[j] = (col < 8 && px < 32) ? (unsigned)(px * (TA_COUT * 2) + col * 16) : 0xFFFFFFF0u;
  }
  auto y_dma = [&](int j, int buf, const TaStep& s) __attribute__((always_inline)) {
    if (SLV_TA_ABL == 2) return;
    if (wave + 4 * j < 5) {
      const unsigned la = lds_base + 2u * TA_XBUF + (unsigned)buf * TA_YBUF + (unsigned)(wave + 4 * j) * 1024u;
      ta_dma16(__builtin_amdgcn_readfirstlane(la),
               (yo[j] != 0xFFFFFFF0u && ypx[j] < s.nr) ? s.row0 * (TA_COUT * 2u) + yo[j] : 0xFFFFFFF0u, ry);
    }
  };

  // ---- fragments: lane (fr, fk) supplies the address of position 4 fk + (fr >> 2) (+16), 4 channels at 4 (fr & 3)
  const int p0 = 4 * fk + (fr >> 2);
  const int xlane = kind * TA_XTILE + p0 * TA_XPB + 8 * (fr & 3);
  const int ylane = p0 * TA_YPB + hf * 64 + 8 * (fr & 3);
  auto rd2 = [&](const unsigned char* lo_p, int hi_off) __attribute__((always_inline)) {
    if (SLV_TA_ABL == 4) return __builtin_bit_cast(bf16x8, (u32x4){(unsigned)(unsigned long)lo_p, 1u, 2u, (unsigned)hi_off});
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(lo_p));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(lo_p + hi_off));
    return tr_pair(lo, hi);
  };
  auto rd_x = [&](int buf, int ci) __attribute__((always_inline)) { return rd2(xlds + buf * TA_XBUF + xlane + ci * 32, 16 * TA_XPB); };
  auto rd_y = [&](int buf, int c) __attribute__((always_inline)) { return rd2(ylds + buf * TA_YBUF + ylane + c * 32, 16 * TA_YPB); };

  f32x4 acc[54];                                        // [co-tile 2][tap 3][ci-tile 9]
#pragma unroll
  for (int m = 0; m < 54; ++m) acc[m] = (f32x4){0.f, 0.f, 0.f, 0.f};
  bf16x8 xf[3][TA_NCI], dyf[3][2];                      // dyf[s % 3]: the dY fragments of step s
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < TA_NCI; ++c) xf[r][c] = __builtin_bit_cast(bf16x8, (u32x4){0u, 0u, 0u, 0u});

  // One step s = frame t of a column: ring slot R = s % 3 (compile time) holds frame t-1, R+1 frame t, R+2 takes frame t+1;
  // register set R holds the y pieces of step s+2 (requested during step s-3).  Slot order: 18 MFMAs of tap -1, 18 of tap 0
  // (their fragments are in registers), 18 of tap +1; behind MFMA m:
  //   m < 9           the two reads of fragment m of frame t+1 (staged during step s-1)
  //   m = 9, 10       the dY fragments of step s+1
  //   m = 11          s_waitcnt vmcnt(2 x requests per step): everything requested during step s-3 has arrived
  //   m = 12 .. 40    (even) the 15 staging items of step s+2 out of register set R
  //   m = 42 .. 44    the three y requests of step s+5 into register set R,  m = 45, 46 its dY DMA
  auto step = [&](int R, int sidx, int xb_rd, int xb_wr, int yb_next, int yb_fill, const TaStep& sf) __attribute__((always_inline)) {
    const int RM = R, R0 = (R + 1) % 3, RP = (R + 2) % 3;
#pragma unroll
    for (int m = 0; m < 54; ++m) {
      const int tap = m / 18, ci = (m % 18) / 2, co = m & 1;
      const int ring = tap == 0 ? RM : (tap == 1 ? R0 : RP);
      if (SLV_TA_ABL == 3) acc[(co * 3 + tap) * TA_NCI + ci][0] += __builtin_bit_cast(float, __builtin_bit_cast(u32x4, dyf[RM][co])[0] ^ __builtin_bit_cast(u32x4, xf[ring][ci])[0]);
      else asm("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[(co * 3 + tap) * TA_NCI + ci]) : "v"(dyf[RM][co]), "v"(xf[ring][ci]));
      if (m < 9) xf[RP][m] = rd_x(xb_rd, m);
      else if (m < 11) dyf[R0][m - 9] = rd_y(yb_next, m - 9);
      else if (m == 11) {
        if (wave == 0) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");       // wave 0: 3 + 2 requests per step
        else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");                  // the others: 3 + 1
      } else if (m >= 12 && m < 42 && ((m - 12) & 1) == 0) x_item(R, (m - 12) / 10, ((m - 12) >> 1) % 5, xb_wr);
      else if (m >= 42 && m < 45) x_load(R, m - 42, sf);
      else if (m >= 45 && m < 47) y_dma(m - 45, yb_fill, sf);
      __builtin_amdgcn_sched_barrier(0);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  };

  // ---- pipeline head: X(0) -> X buffer 0, X(1) -> X buffer 1 (staged right away), y of the steps 2, 3, 4 into the register
  // sets 0, 1, 2, dY of the steps 0 .. 4 into the ring's tiles 0 .. 4; then the fragments of frame 0 into ring slot 1
  if (nsteps > 0) {
    constexpr int S0 = 0, S1 = 1, S2 = 2;
    const TaStep s0 = next_step();
#pragma unroll
    for (int i = 0; i < 3; ++i) x_load(S0, i, s0);
#pragma unroll
    for (int j = 0; j < 2; ++j) y_dma(j, 0, s0);
    const TaStep s1 = next_step();
#pragma unroll
    for (int i = 0; i < 3; ++i) x_load(S1, i, s1);
#pragma unroll
    for (int j = 0; j < 2; ++j) y_dma(j, 1, s1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);            // (the asm orders memory operations only: pin the register uses behind it)
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int d = 0; d < 5; ++d) x_item(S0, i, d, 0);
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int d = 0; d < 5; ++d) x_item(S1, i, d, 1);
    const TaStep s2 = next_step();
#pragma unroll
    for (int i = 0; i < 3; ++i) x_load(S0, i, s2);
#pragma unroll
    for (int j = 0; j < 2; ++j) y_dma(j, 2, s2);
    const TaStep s3 = next_step();
#pragma unroll
    for (int i = 0; i < 3; ++i) x_load(S1, i, s3);
#pragma unroll
    for (int j = 0; j < 2; ++j) y_dma(j, 3, s3);
    const TaStep s4 = next_step();
#pragma unroll
    for (int i = 0; i < 3; ++i) x_load(S2, i, s4);
#pragma unroll
    for (int j = 0; j < 2; ++j) y_dma(j, 4, s4);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int c = 0; c < TA_NCI; ++c) xf[1][c] = rd_x(0, c);          // frame 0 -> slot R0 of step 0
#pragma unroll
    for (int c = 0; c < 2; ++c) dyf[0][c] = rd_y(0, c);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                                      // X buffer 0 is free for X(2)
  }
  // step s: X(s+1) fragments from X buffer (s+1) & 1; X(s+2) staged into X buffer s & 1; dY(s+1) fragments from dY tile
  // (s+1) % 6; dY(s+5) by DMA into tile (s+5) % 6
  int yn = 1, yf = 5;                                  // (the virtual frame's dY tile is zeros: its MFMAs add nothing)
  for (int s = 0; s < nsteps; s += 3) {
#define TA_STEP(R_, S_)                                                                                     \
    if ((S_) < nsteps) {                                                                                    \
      const TaStep sf = next_step();                                                                        \
      step(R_, (S_), ((S_) + 1) & 1, (S_) & 1, yn, yf, sf);          \
      yn = yn == TA_NYB - 1 ? 0 : yn + 1;                                                                   \
      yf = yf == TA_NYB - 1 ? 0 : yf + 1;                                                                   \
    }
    TA_STEP(0, s)
    TA_STEP(1, s + 1)
    TA_STEP(2, s + 2)
#undef TA_STEP
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                   // (requests past the end of the sequence)

  // ---- this workgroup's partial: part[kind][wg][co][tap * 160 + ci]; C/D layout: column (ci) = lane & 15, rows (co) 4 fk + r
  asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");
  float* pw = part + (size_t)kind * kind_stride + (size_t)blockIdx.x * (TA_COUT * 3 * TA_CINP);
#pragma unroll
  for (int m = 0; m < 54; ++m) {
    const int co = m / 27, tap = (m / 9) % 3, ci = m % 9;
#pragma unroll
    for (int r = 0; r < 4; ++r)
      pw[(size_t)((hf * 2 + co) * 16 + 4 * fk + r) * (3 * TA_CINP) + tap * TA_CINP + ci * 16 + fr] = acc[m][r];
  }
}

static bool tacc_enabled() {
  static const bool on = []() {
    const char* e = getenv("SELAVI_CL16_WGTACC");
    return !(e && e[0] == '0');
  }();
  return on;
}

int wgrad_tacc_blocks() {
  static const int blocks = []() {
    const char* e = getenv("SELAVI_CL16_WGTACC_BLOCKS");
    const int b = e ? atoi(e) : 256;
    return b < 1 ? 1 : b;
  }();
  return blocks;
}

// Conv3d(144 -> 64, (3,1,1), stride 1, padding (1,0,0)): enough columns to keep every workgroup busy
bool wgrad_tacc_applies(const ClWgradT& g) {
  if (!tacc_enabled()) return false;
  if (g.Cin_p != TA_CINP || g.Cin != 144 || g.Cout_p != TA_COUT) return false;
  if ((long long)g.N * g.PB < 4LL * wgrad_tacc_blocks()) return false;
  return true;
}

int wgrad_tacc_launch(const ClWgradT& g, const void* dy, const void* x, const float* in_ss, float* part, size_t kind_stride,
                      hipStream_t st) {
  static bool attr_set = false;
  if (!attr_set) {
    SLV_HIP(hipFuncSetAttribute((const void*)cl16_wgrad_tacc_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_set = true;
  }
  hipLaunchKernelGGL(cl16_wgrad_tacc_kernel, dim3(wgrad_tacc_blocks()), dim3(256), TA_LDS, st, (const unsigned short*)dy,
                     (const unsigned short*)x, in_ss, part, kind_stride, g.N, g.T, g.HW, g.PB, g.Cin);
  return launch_check("slv_cl16_wgrad_bnr");
}

}  // namespace slv
