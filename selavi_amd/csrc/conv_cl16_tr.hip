// 16-bit MFMA path: the stride-1 temporal (3,1,1) convolutions of the narrow layers (stem, layer 1: a quarter of the
// trunk's activation bytes), forward and backward data, with the WEIGHTS RESIDENT IN REGISTERS.
//
// These convs are tiny GEMMs per position (K = 3 x 160 -> 64 channels, or 3 x 64 -> 144) over 10^7 positions: by
// arithmetic intensity (~130 FLOP/B) they are bound by HBM, and what the tile kernels (csrc/conv_cl16.hip) spend their
// time on is everything BUT the bytes -- re-streaming the same 46-55 KB of weights through LDS for every 128 positions
// (LDS-DMA issue, a barrier per K-step), position decodes, three fetches of every activation row (one per tap).
// Here the roles are turned around:
//   * a workgroup is ONE wave; it loads the whole weight tensor once, as MFMA A fragments, into registers
//     (MT x KC x 3 fragments = 216-240 VGPRs of the wave's 512: one wave per SIMD) and keeps it for its lifetime;
//   * the wave walks COLUMNS: 32 pixels of one clip, frame by frame.  A ring of three frames lives in the wave's own
//     LDS region; frame t+2 is fetched (coalesced 16-byte pieces, BatchNorm + ReLU applied once per element on the
//     way) while frame t is computed from the ring, so every activation row is read from memory exactly once and the
//     three taps are three reads of the same LDS rows;
//   * nothing is shared between waves: no barrier, no LDS-DMA bookkeeping, no block-level phases -- the only
//     synchronisation is the wave's own instruction order;
//   * B fragments come out of LDS (one ds_read_b128 feeds MT MFMAs), the output tile goes through LDS once to be
//     stored in whole channel rows, BatchNorm statistics are taken from that tile on the matrix cores
//     (cl16.hpp:wave_tile_stats' scheme) and accumulated in registers over the column: one partial per column.
// Frames outside the clip (temporal zero padding) are skipped MFMAs, not zero rows.
#include "cl16.hpp"
#include "../../include/selavi_hip.h"
#include <utility>

namespace slv {

constexpr int TR_PX = 32;                       // pixels per wave tile: two 16-position MFMA fragments
constexpr int TR_CHS = TR_PX * 64 + 64;         // bytes from a chunk of a ring slot to the next: + 16 banks, so that the four
                                                // chunks one quarter of a staging store touches do not share their banks
                                                // (2 048-byte chunks: every ds_write_b128 of a frame was a 4-way conflict)

// acc += A * B with the A fragment held in the ACCUMULATOR half of the register file.  Left to itself hipcc parks the
// 216-240 registers of resident weights in AGPRs as a spill area and copies every fragment back with four
// v_accvgpr_read (+ s_nop) in front of each MFMA pair, through ONE VGPR quad -- 2 VALU per MFMA and a serial chain.
// An "a" operand makes the MFMA read the fragment where it lives.  (Inline asm: the compiler neither pads the MFMA's
// hazards nor knows its latency; tr_mfma_settle() below is the wait between the last MFMA and the first VALU read of an
// accumulator.  The B operand comes from ds_read -- the compiler's own lgkmcnt wait covers it -- never from a VALU.)
#ifndef SLV_TR_INTERLEAVE
#define SLV_TR_INTERLEAVE 0      // 0: BatchNorm + ReLU of the staged frame after the step's MFMAs; 1: between the MFMA groups
#endif                           // (compiler-placed); 2: pinned there with sched_barrier.  Measured (layer-1 temporal forward,
                                 // 64 clips): see profiles/r03_notes.md
#ifndef SLV_TR_ASM_MFMA
#define SLV_TR_ASM_MFMA 1
#endif
__device__ __forceinline__ void tr_mfma(f32x4& acc, const bf16x8& a, const bf16x8& b) {
#if SLV_TR_ASM_MFMA
  asm("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc) : "a"(a), "v"(b));
#else
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc, 0, 0, 0);
#endif
}
// acc = A * B (first MFMA of a step: no accumulator to clear)
__device__ __forceinline__ void tr_mfma0(f32x4& acc, const bf16x8& a, const bf16x8& b) {
#if SLV_TR_ASM_MFMA
  asm("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=v"(acc) : "a"(a), "v"(b));
#else
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, (f32x4){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
#endif
}
// one fragment from LDS, invisible to the compiler's scheduler and wait counting (see the step's MFMA section)
template <int OFF>
__device__ __forceinline__ void tr_lds_read(bf16x8& d, unsigned addr) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(OFF));
}
template <class F, int... I>
__device__ __forceinline__ void tr_for_seq(std::integer_sequence<int, I...>, F&& f) {
  (f(std::integral_constant<int, I>{}), ...);
}
__device__ __forceinline__ void tr_mfma_settle() {
#if SLV_TR_ASM_MFMA
  asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");      // 8-pass XDL result -> VALU read: 11+ wait states
#endif
}

#ifdef SLV_TR_TRACE      // timing only (tools/tr_trace.py): s_memtime ticks per section of a step, summed over the wave's steps, written
#define TR_STAMP(i) do { const unsigned long long t_ = __builtin_readcyclecounter(); trc[i] += t_ - tlast; tlast = t_; } while (0)
#else                    // over the head of y at the end of the kernel
#define TR_STAMP(i) do { } while (0)
#endif

// MT: 16-row tiles of output channels (all of them: Mrows == 16 MT); KC: 32-channel chunks of the input (Cin_p / 32).
// PRO 1: rows are read as relu(x * s + h).  EPI 0: y = acc -> bf16.  EPI 1: + per-channel sum / sum of squares of the
// rounded outputs, one partial per column: stat_sum / stat_sq [Cout][ncol].  EPI 3 (backward data whose BatchNorm-backward
// coefficients are known BEFORE it runs -- they come out of the weight gradient, csrc/wgrad_cl16_t2.hip): the stored
// value is A1 * mask * g + A2 + A3 * x with g = the rounded gradient, x = ax (raw output of the layer this conv read, same
// positions), mask = [x s + h > 0], ab5 = {s, h, A1, A2, A3}[Cout] -- what slv_cl16_bn_bwd_apply would make of the stored g
// in a pass of its own (read g, read x, write), bit for bit.
template <int MT, int KC, int PRO, int EPI>
__global__ __launch_bounds__(64, 1) void conv_cl16_tr_kernel(const unsigned short* __restrict__ x,
                                                            const unsigned short* __restrict__ wl,
                                                            unsigned short* __restrict__ y,
                                                            const float* __restrict__ in_ss,
                                                            float* __restrict__ stat_sum, float* __restrict__ stat_sq,
                                                            ClConv g, int ncol, int pbn,
                                                            const unsigned short* __restrict__ ax,
                                                            const float* __restrict__ ab5) {
  constexpr int PPR = KC * 4;                         // 16-byte pieces per input row
  constexpr int RPI = 64 / PPR;                       // input rows one load instruction covers
  constexpr int NIT = (TR_PX + RPI - 1) / RPI;
  constexpr int SLOT = KC * TR_CHS + 1024;             // bytes of one frame in the ring: [chunk][pixel][64 B] + 1 KiB where the
                                                      // idle lanes' pieces go (no branches, one address form for every lane)
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  unsigned char* const ring = lds;                    // 3 x SLOT
  typedef __attribute__((address_space(3))) void* tr_lds_void;
  const unsigned ring_a = (unsigned)(unsigned long)(tr_lds_void)lds;
  unsigned char* const ost = lds + 3 * SLOT;          // [TR_PX][orow]
  const int lane = threadIdx.x;
  const int fr = lane & 15, fk = lane >> 4;
  const int T = g.Ti, HW = g.Hi * g.Wi;
  constexpr int COUTP = ((MT * 16 + 31) / 32) * 32;   // (Cout_p == COUTP: cl16_tr_applies)
  constexpr int orow = COUTP * 2 + 16;                // bytes per pixel row of the output stage (+16: banks)
  const unsigned in_row = (unsigned)g.Cin_p * 2u, out_row = (unsigned)g.Cout_p * 2u;
  const unsigned Ptot = (unsigned)g.N * T * HW;
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, (int)(Ptot * in_row), 0x00020000);
  const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc((void*)y, 0, (int)(Ptot * out_row), 0x00020000);

  // ---- the weights: A fragments of every (tap, chunk, cout tile), resident for the kernel's lifetime
  int dt[3], slab[3];
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    dt[j] = (g.tap[j] & 15) - 8 + g.bot;
    slab[j] = g.tap[j] >> 12;
  }
  bf16x8 A[3][KC][MT];
#pragma unroll
  for (int j = 0; j < 3; ++j)
#pragma unroll
    for (int c = 0; c < KC; ++c)
#pragma unroll
      for (int i = 0; i < MT; ++i)
        A[j][c][i] = *(const bf16x8*)(wl + ((size_t)((slab[j] * KC + c) * g.Mrows + i * 16 + fr) * 32 + fk * 8));

  // ---- this lane's part of a frame load: the SAME 16-byte channel piece of rows lr, lr + RPI, ...
  const int piece = lane % PPR, lr = lane / PPR;
  const bool lact = lane < RPI * PPR;
  float ps[8], ph[8];
  if constexpr (PRO == 1) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int c = piece * 8 + e;
      const bool ok = lact && c < g.Cin;
      ps[e] = ok ? in_ss[c] : 0.f;
      ph[e] = ok ? in_ss[g.Cin + c] : 0.f;
    }
  }
  // LDS position of (row r, piece): chunk = piece / 4, slot (piece & 3) ^ swz(r)
  const int wchunk = piece >> 2, wq = piece & 3;
  // fragment reads: lane (fr, fk) reads row n * 16 + fr, slot fk ^ swz(fr)  (swz(n * 16 + fr) == swz(fr))
  const int boff = fr * 64 + ((fk ^ cl_swz(fr)) << 4);
  // output store: the same 16-byte piece of rows olr, olr + ORPI, ...
  constexpr int oppr = COUTP >> 3, orpi = 64 / oppr, ONIT = (TR_PX + orpi - 1) / orpi;
  const int opiece = lane % oppr, olr = lane / oppr;
  const bool oact = lane < orpi * oppr;
  unsigned ooff[ONIT];                                // (per lane and store instruction, as loff / sto / lrow above)
  int olds[ONIT], orw[ONIT];
#pragma unroll
  for (int q = 0; q < ONIT; ++q) {
    const int r = q * orpi + olr;
    const bool in = oact && r < TR_PX;
    ooff[q] = (unsigned)r * (COUTP * 2u) + opiece * 16u;
    orw[q] = in ? r : 0x7FFFFFFF;
    olds[q] = (r < TR_PX ? r : 0) * (COUTP * 2 + 16) + opiece * 16;
  }
  float e_s[8], e_h[8], e_a1[8], e_a2[8], e_a3[8];
  if constexpr (EPI == 3) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int c = opiece * 8 + e;
      const bool ok = oact && c < g.Cout;
      e_s[e] = ok ? ab5[c] : 0.f;
      e_h[e] = ok ? ab5[g.Cout + c] : 0.f;
      e_a1[e] = ok ? ab5[2 * g.Cout + c] : 0.f;
      e_a2[e] = ok ? ab5[3 * g.Cout + c] : 0.f;
      e_a3[e] = ok ? ab5[4 * g.Cout + c] : 0.f;
    }
  }
  const __amdgpu_buffer_rsrc_t rax = __builtin_amdgcn_make_buffer_rsrc((void*)(EPI == 3 ? ax : x), 0, (int)(Ptot * out_row), 0x00020000);

  // zero the output stage once: the channel pieces beyond the accumulator tiles (Cout_p > 16 MT) stay zero
  for (int i = lane * 16; i < TR_PX * orow; i += 64 * 16) *(u32x4*)(ost + i) = (u32x4){0u, 0u, 0u, 0u};

  // ---- the wave's work: columns blockIdx.x, + gridDim.x, ...; flattened into STEPS (column k, frame t) so that the
  // fetch pipeline runs across column borders.  Step s lives in ring slot s % 3; at step s the frame of step s + 3 is
  // requested into one register set while the other set (step s + 2, requested a step earlier) gets its BatchNorm +
  // ReLU between this step's MFMA groups and is written to the slot of step s - 1 once the MFMAs have read it.
  const int ncols = blockIdx.x < (unsigned)ncol ? (ncol - 1 - (int)blockIdx.x) / (int)gridDim.x + 1 : 0;
  const int S = ncols * T;
  struct Cur {
    int k, t;                                         // column index of this wave, frame
    int px0;
    unsigned pos0;                                    // position of (frame 0, first pixel of the column)
  };
  auto col_of = [&](Cur& c) __attribute__((always_inline)) {
    const int col = blockIdx.x + c.k * gridDim.x;
    const int n = col / pbn, pb = col - n * pbn;
    // (a step beyond the wave's last column has no pixel at all: px0 = HW turns every lane's bounds test off -- as a
    //  PER-LANE condition; a wave-uniform "is this step live" in front of the loads makes hipcc branch around each load
    //  and drain vmcnt(0) between them)
    c.px0 = c.k < ncols ? pb * TR_PX : HW;
    c.pos0 = (unsigned)n * T * HW + c.px0;
  };
  auto advance = [&](Cur& c) __attribute__((always_inline)) {
    if (++c.t == T) {
      c.t = 0;
      ++c.k;
      col_of(c);
    }
  };
  // per lane and load instruction (loop invariants; round 5 -- the trace of tools/tr_trace.py showed 870 cycles per step in
  // front of the MFMAs for 11 requests whose offsets were rebuilt from the position every time, 1 340 behind them for the LDS
  // stores with their swizzle, 1 270 for four dependent read-then-store rounds of the output rows): the byte offset of the
  // piece relative to the frame's first pixel of the column, its place in a ring slot, the row it belongs to
  unsigned loff[NIT];
  int sto[NIT], lrow[NIT];
#pragma unroll
  for (int i = 0; i < NIT; ++i) {
    const int r = lr + RPI * i;
    const bool in = lact && r < TR_PX;
    loff[i] = (unsigned)r * in_row + piece * 16u;
    lrow[i] = in ? r : 0x7FFFFFFF;                    // (idle lanes: a row no frame has)
    sto[i] = in ? wchunk * TR_CHS + r * 64 + ((wq ^ cl_swz(r)) << 4) : KC * TR_CHS + lane * 16;
  }
  auto load_frame = [&](const Cur& c, u32x4* st, unsigned& stv) __attribute__((always_inline)) {
    stv = 0;
    // (the out-of-range marker does not survive a scalar offset -- the sum wraps back into the buffer: the base goes into
    // the vector offset, three VALU instructions per request instead of eight)
    const unsigned base = (c.pos0 + (unsigned)c.t * HW) * in_row;
    const int left = HW - c.px0;                      // pixels of the frame from the column's first one (<= 0: no such step)
#pragma unroll
    for (int i = 0; i < NIT; ++i)
      st[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, lrow[i] < left ? base + loff[i] : 0xFFFFFFF0u, 0, CL_NT));
  };
  auto affine_piece = [&](u32x4* st, unsigned stv, int i) __attribute__((always_inline)) {
    // (rows beyond the frame's pixels -- the ragged last block -- become relu(h) instead of zero: their outputs are
    //  never stored, and the statistics zero those rows of the output tile first)
    if constexpr (PRO == 1) st[i] = affine_relu8(st[i], ps, ph);
    (void)stv;
  };
  auto store_frame = [&](int slot, const u32x4* st) __attribute__((always_inline)) {
    unsigned char* dst = ring + slot * SLOT;
#pragma unroll
    for (int i = 0; i < NIT; ++i) {
      *(u32x4*)(dst + sto[i]) = st[i];
    }
  };
  u32x4 ra[NIT], rb[NIT];
  unsigned va = 0, vb = 0;
  Cur pf = {0, 0, 0, 0u};                             // the step being FETCHED
  if (S > 0) col_of(pf);
  // pipeline head: steps 0 and 1 into their slots, step 2 into register set a
  load_frame(pf, ra, va);
  advance(pf);
  load_frame(pf, rb, vb);
  advance(pf);
#pragma unroll
  for (int i = 0; i < NIT; ++i) affine_piece(ra, va, i);
  store_frame(0, ra);
#pragma unroll
  for (int i = 0; i < NIT; ++i) affine_piece(rb, vb, i);
  store_frame(1, rb);
  load_frame(pf, ra, va);
  advance(pf);
  Cur cc = {0, 0, 0, 0u};                             // the step being COMPUTED
  if (S > 0) col_of(cc);
  int slot = 0;                                       // ring slot of the computed step
  float accS[MT], accQ[MT];
#pragma unroll
  for (int i = 0; i < MT; ++i) accS[i] = accQ[i] = 0.f;
  // one step: `cur` = register set holding step s + 2 (requested one step ago), `nxt` = the set step s + 3 goes to
#ifdef SLV_TR_TRACE
  unsigned long long trc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, tlast = __builtin_readcyclecounter();
#endif
  auto step = [&](u32x4* cur, unsigned& vcur, u32x4* nxt, unsigned& vnxt) __attribute__((always_inline)) {
    load_frame(pf, nxt, vnxt);                        // two frames in flight behind this step's MFMAs
    advance(pf);
    // EPI 3: the x pieces this step's epilogue needs (160 stored channels: 3 rows per instruction, 11 instructions) are
    // requested HERE, in front of the step's MFMAs (requested inside the store loop they cost one memory round trip each:
    // 3.83 ms per launch)
    constexpr int E3_NOIT = ONIT;
    u32x4 yv[EPI == 3 ? E3_NOIT : 1];
    const unsigned obase = (cc.pos0 + (unsigned)cc.t * HW) * out_row;   // the step's first output row
    const int oleft = HW - cc.px0;
    if constexpr (EPI == 3) {
#pragma unroll
      for (int q = 0; q < E3_NOIT; ++q)
        yv[q] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rax, orw[q] < oleft ? obase + ooff[q] : 0xFFFFFFF0u, 0, 0));
    }
    const int sprev = slot == 0 ? 2 : slot - 1, snext = slot == 2 ? 0 : slot + 1;
    TR_STAMP(0);                                      // requests issued
    f32x4 acc[MT][2];
    int pi = 0;                                       // next staged piece to get its BatchNorm + ReLU
    if constexpr (MT * KC * 3 <= 24) {
      // (the stem's 45 <-> 64 kernels: two waves per SIMD cover each other's fragment reads; the hand-placed reads below
      //  measured 10 % slower there -- 0.171 -> 0.189 ms with the prologue -- so the compiler keeps this form)
#pragma unroll
      for (int jj = 0; jj < 3; ++jj) {
        const int j = jj == 0 ? 1 : jj == 1 ? 0 : 2;    // the centre tap (always inside the clip) first: it initialises
        const int f = cc.t + dt[j];
        const bool live = f >= 0 && f < T;              // temporal zero padding: nothing to add
        const unsigned char* src = ring + (dt[j] < 0 ? sprev : dt[j] > 0 ? snext : slot) * SLOT + boff;
#pragma unroll
        for (int c = 0; c < KC; ++c) {
          if (live) {
            const bf16x8 b0 = *(const bf16x8*)(src + c * TR_CHS);
            const bf16x8 b1 = *(const bf16x8*)(src + c * TR_CHS + 16 * 64);
#pragma unroll
            for (int i = 0; i < MT; ++i) {
              if (jj == 0 && c == 0) {
                tr_mfma0(acc[i][0], A[j][c][i], b0);
                tr_mfma0(acc[i][1], A[j][c][i], b1);
              } else {
                tr_mfma(acc[i][0], A[j][c][i], b0);
                tr_mfma(acc[i][1], A[j][c][i], b1);
              }
            }
          }
#if SLV_TR_INTERLEAVE >= 1      // the VALU work of the staged frame in the shadow of the MFMA groups
          constexpr int GROUPS = 3 * KC, PER = (NIT + GROUPS - 1) / GROUPS;
#pragma unroll
          for (int q = 0; q < PER; ++q)
            if (pi < NIT) affine_piece(cur, vcur, pi++);
#if SLV_TR_INTERLEAVE >= 2
          __builtin_amdgcn_sched_barrier(0);
#endif
#endif
        }
      }
    } else {
      // 3 KC groups (tap, chunk) of 2 fragment reads + 2 MT MFMAs.  The reads of group g + 1 are issued IN FRONT of the MFMAs of
      // group g (two register sets), unconditionally -- only the MFMAs of a tap outside the clip (temporal zero padding) are
      // skipped.  Left to the compiler, the reads sat inside the per-group "is this tap live" block, each group began with a full
      // LDS round trip and the 120 MFMAs of a step took 3 175 cycles instead of 2 160 (tools/tr_trace.py, round 5).  Inline asm:
      // the compiler must neither sink the reads into the conditional block nor wait for them early; the wait is a full drain
      // (lgkmcnt also counts scalar loads the compiler may have in flight) placed BEFORE the next group's reads are issued.
      unsigned sa[3];                                   // LDS byte address of this lane's fragment rows in the tap's ring slot
      bool lv[3];
#pragma unroll
      for (int jj = 0; jj < 3; ++jj) {
        const int j = jj == 0 ? 1 : jj == 1 ? 0 : 2;    // the centre tap (always inside the clip) first: it initialises
        const int f = cc.t + dt[j];
        lv[jj] = f >= 0 && f < T;                       // temporal zero padding: nothing to add
        sa[jj] = ring_a + (unsigned)((dt[j] < 0 ? sprev : dt[j] > 0 ? snext : slot) * SLOT + boff);
      }
      bf16x8 bq[2][2];
      tr_lds_read<0>(bq[0][0], sa[0]);
      tr_lds_read<16 * 64>(bq[0][1], sa[0]);
      tr_for_seq(std::make_integer_sequence<int, 3 * KC>{}, [&](auto G) __attribute__((always_inline)) {
        constexpr int gq = decltype(G)::value, jj = gq / KC, c = gq % KC;
        const int j = jj == 0 ? 1 : jj == 1 ? 0 : 2;
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(bq[gq & 1][0]), "+v"(bq[gq & 1][1]));      // group gq's fragments have landed
        if constexpr (gq + 1 < 3 * KC) {
          constexpr int jn = (gq + 1) / KC, cn = (gq + 1) % KC;
          tr_lds_read<cn * TR_CHS>(bq[(gq + 1) & 1][0], sa[jn]);
          tr_lds_read<cn * TR_CHS + 16 * 64>(bq[(gq + 1) & 1][1], sa[jn]);
        }
        if (lv[jj]) {
#pragma unroll
          for (int i = 0; i < MT; ++i) {
            if constexpr (gq == 0) {
              tr_mfma0(acc[i][0], A[j][c][i], bq[0][0]);
              tr_mfma0(acc[i][1], A[j][c][i], bq[0][1]);
            } else {
              tr_mfma(acc[i][0], A[j][c][i], bq[gq & 1][0]);
              tr_mfma(acc[i][1], A[j][c][i], bq[gq & 1][1]);
            }
          }
        }
#if SLV_TR_INTERLEAVE >= 1      // the VALU work of the staged frame in the shadow of the MFMA groups
        constexpr int GROUPS = 3 * KC, PER = (NIT + GROUPS - 1) / GROUPS;
#pragma unroll
        for (int q = 0; q < PER; ++q)
          if (pi < NIT) affine_piece(cur, vcur, pi++);
#if SLV_TR_INTERLEAVE >= 2
        __builtin_amdgcn_sched_barrier(0);
#endif
#endif
      });
    }
    TR_STAMP(1);                                      // the MFMAs (and the fragment reads they wait for)
#pragma unroll
    for (; pi < NIT; ++pi) affine_piece(cur, vcur, pi);
    TR_STAMP(2);                                      // the wait for the staged frame + its BatchNorm + ReLU
    // ---- epilogue of the step: bf16 tile [pixel][cout] through LDS
    tr_mfma_settle();
#if SLV_TR_ASM_MFMA
#pragma unroll
    for (int i = 0; i < MT; ++i) mfma_pin(acc[i][0]), mfma_pin(acc[i][1]);
#endif
    TR_STAMP(8);                                      // (trace: the settle nops)
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int nn = 0; nn < 2; ++nn) {
        const unsigned lo = pack_bf2(acc[i][nn][0], acc[i][nn][1]), hi = pack_bf2(acc[i][nn][2], acc[i][nn][3]);
        *(uint2*)(ost + (nn * 16 + fr) * orow + (i * 16 + fk * 4) * 2) = make_uint2(lo, hi);
      }
    TR_STAMP(9);                                      // (trace: rounding + the output tile's LDS writes)
    store_frame(sprev, cur);                          // step s + 2 takes the slot of step s - 1: no longer read
    TR_STAMP(3);                                      // output tile + staged frame into LDS
    if constexpr (EPI == 1) {                         // statistics of the 32 rounded rows on the matrix cores
      if (cc.px0 + TR_PX > HW) {                      // ragged last block of a frame: rows without a pixel count as zero
        const int valid = HW - cc.px0;
        for (int i = lane * 16; i < TR_PX * orow; i += 64 * 16)
          if (i / orow >= valid) *(u32x4*)(ost + i) = (u32x4){0u, 0u, 0u, 0u};
      }
      wave_rows32_stats_acc<MT>(ost, orow, lane, accS, accQ);
    }
    TR_STAMP(4);                                      // statistics
    {
      if constexpr (EPI == 3) {
#pragma unroll
        for (int q = 0; q < E3_NOIT; ++q) {
          u32x4 v = *(const u32x4*)(ost + olds[q]);
          const u32x4 xv = yv[q];
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            float o2[2];
#pragma unroll
            for (int e = 0; e < 2; ++e) {
              const int k = 2 * i + e;
              const float xx = e ? bf_hi(xv[i]) : bf_lo(xv[i]);
              float gg = e ? bf_hi(v[i]) : bf_lo(v[i]);
              if (!(bn_affine(xx, e_s[k], e_h[k]) > 0.f)) gg = 0.f;
              o2[e] = bn_bwd_apply1(gg, xx, e_a1[k], e_a2[k], e_a3[k]);      // (cl16_bn_bwd_apply_kernel's)
            }
            v[i] = pack_bf2(o2[0], o2[1]);
          }
          __builtin_amdgcn_raw_buffer_store_b128(v, ry, orw[q] < oleft ? obase + ooff[q] : 0xFFFFFFF0u, 0, CL_NT);
        }
      } else {
        u32x4 v[ONIT];                                // all reads of the output tile, then all stores
#pragma unroll
        for (int q = 0; q < ONIT; ++q) v[q] = *(const u32x4*)(ost + olds[q]);
#pragma unroll
        for (int q = 0; q < ONIT; ++q)                // (out-of-range offsets drop the store: no branches)
          __builtin_amdgcn_raw_buffer_store_b128(v[q], ry, orw[q] < oleft ? obase + ooff[q] : 0xFFFFFFF0u, 0, CL_NT);
      }
    }
    if constexpr (EPI == 1) {
      if (cc.t == T - 1) {                            // the column is complete: its partial sums
        const int col = blockIdx.x + cc.k * gridDim.x;
#pragma unroll
        for (int i = 0; i < MT; ++i) {
          const int c = i * 16 + fr;
          if (c < g.Cout) {
            if (fk == 0) stat_sum[(size_t)c * ncol + col] = accS[i];
            if (fk == (fr >> 2)) stat_sq[(size_t)c * ncol + col] = accQ[i];
          }
          accS[i] = accQ[i] = 0.f;
        }
      }
    }
    advance(cc);
    slot = snext;
    TR_STAMP(5);                                      // output rows to memory (EPI 3: + the BatchNorm-backward apply)
  };
  for (int s = 0; s < S; s += 2) {
    step(ra, va, rb, vb);
    if (s + 1 < S) step(rb, vb, ra, va);
  }
#ifdef SLV_TR_TRACE
  if (lane == 0) {
    unsigned long long* o_ = (unsigned long long*)y + blockIdx.x * 10;
    for (int i_ = 0; i_ < 10; ++i_) o_[i_] = i_ == 7 ? (unsigned long long)S : trc[i_];
  }
#endif
}

static bool tr_enabled() {
  static const bool enabled = []() {
    const char* e = getenv("SELAVI_CL16_TR");
    return !(e && e[0] == '0');
  }();
  return enabled;
}

// (MT, KC) instantiated below
static bool tr_shape(int mt, int kc) { return (mt == 4 && (kc == 5 || kc == 2)) || (mt == 9 && kc == 2); }

// Does this launch fit?  Three taps along t at offsets -1 / 0 / +1 (in either order), stride 1, lattice == input ==
// output positions, every output channel in one tile group, weights that fit the register file.
bool cl16_tr_applies(const ClConv& g) {
  if (!tr_enabled() || g.ntaps != 3) return false;
  if (g.Lt != g.Ti || g.Lh != g.Hi || g.Lw != g.Wi || g.To != g.Ti || g.Ho != g.Hi || g.Wo != g.Wi) return false;
  if (g.bmt != 1 || g.bmh != 1 || g.bmw != 1 || g.omt != 1 || g.omh != 1 || g.omw != 1 || g.oot || g.ooh || g.oow) return false;
  int seen = 0;
  for (int t = 0; t < 3; ++t) {
    const int d = (g.tap[t] & 15) - 8 + g.bot, dh = ((g.tap[t] >> 4) & 15) - 8 + g.boh, dw = ((g.tap[t] >> 8) & 15) - 8 + g.bow;
    if (dh != 0 || dw != 0 || d < -1 || d > 1) return false;
    seen |= 1 << (d + 1);
  }
  if (seen != 7) return false;
  {
    static const int only_kc = []() {                 // diagnostic: restrict the kernel to one channel-chunk count
      const char* e = getenv("SELAVI_CL16_TR_ONLY_KC");
      return e ? atoi(e) : 0;
    }();
    if (only_kc && g.Cin_p / 32 != only_kc) return false;
  }
  if ((g.Mrows & 15) || !tr_shape(g.Mrows / 16, g.Cin_p / 32) || g.Cout_p != ((g.Mrows + 31) / 32) * 32) return false;
  if ((long long)g.N * g.Ti * g.Hi * g.Wi * (g.Cin_p > g.Cout_p ? g.Cin_p : g.Cout_p) * 2 >= 0xFFFFFFF0LL) return false;
  return true;
}

int cl16_tr_columns(const ClConv& g) { return g.N * ((g.Hi * g.Wi + TR_PX - 1) / TR_PX); }
// forward launches (taps in ascending order of their offset) own the statistics-partial layout: one slot per column
// (slv_cl16_conv_nblk); backward-data launches keep the tile kernel's slot count, whose fused BatchNorm-backward sums
// (EPI 2) this kernel does not have -- such a launch goes to the tile kernel
bool cl16_tr_forward(const ClConv& g) { return (g.tap[0] & 15) - 8 + g.bot < 0; }

template <int MT, int KC, int PRO, int EPI>
static int tr_launch_one(const ClConv& g, const void* x, const void* wl, void* y, const float* in_ss, float* stat_sum,
                         float* stat_sq, hipStream_t st, const void* ax = nullptr, const float* ab5 = nullptr) {
  const size_t lds = 3 * ((size_t)KC * TR_CHS + 1024) + (size_t)TR_PX * (g.Cout_p * 2 + 16);
  static bool attr_set = false;
  if (!attr_set) {
    SLV_HIP(hipFuncSetAttribute((const void*)conv_cl16_tr_kernel<MT, KC, PRO, EPI>,
                                hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
    attr_set = true;
  }
  const int pbn = (g.Hi * g.Wi + TR_PX - 1) / TR_PX, ncol = g.N * pbn;
  // resident waves: 256 CUs x 4 SIMDs x (2 where the weights leave room for a second wave per SIMD)
  static const int waves_env = []() {
    const char* e = getenv("SELAVI_CL16_TR_WAVES");
    return e ? atoi(e) : 0;
  }();
  const int waves = waves_env > 0 ? waves_env : (MT * KC * 3 <= 24 ? 2048 : 1024);
  const int grid = ncol < waves ? ncol : waves;
  hipLaunchKernelGGL((conv_cl16_tr_kernel<MT, KC, PRO, EPI>), dim3(grid), dim3(64), lds, st, (const unsigned short*)x,
                     (const unsigned short*)wl, (unsigned short*)y, in_ss, stat_sum, stat_sq, g, ncol, pbn,
                     (const unsigned short*)ax, ab5);
  return 0;
}

// backward data of the layer-1 temporal conv (64 -> 144 stored channels) with the BatchNorm-backward apply of the layer in
// front folded into its epilogue (EPI 3)
bool cl16_tr_dgrad_apply_ok(const ClConv& g) {
  return cl16_tr_applies(g) && !cl16_tr_forward(g) && g.Mrows / 16 == 9 && g.Cin_p / 32 == 2;
}

int cl16_tr_dgrad_apply(const ClConv& g, const void* x, const void* wl, void* y, const void* ax, const float* ab5, hipStream_t st) {
  if (!cl16_tr_dgrad_apply_ok(g)) return 0;
  int rc = tr_launch_one<9, 2, 0, 3>(g, x, wl, y, nullptr, nullptr, nullptr, st, ax, ab5);
  if (rc) return rc;
  rc = launch_check("slv_cl16_conv_dgrad_bn_apply");
  return rc ? rc : 1;
}

// returns 1 when the launch was taken, 0 when it does not apply (epilogues this kernel does not have: affine, residual,
// ReLU, fused BatchNorm-backward sums), < 0 on error
int cl16_tr_try(const ClConv& g, const void* x, const void* wl, void* y, const float* in_ss, const float* scale_shift,
                const void* res, int relu, float* stat_sum, float* stat_sq, const ClBnr& bnr, hipStream_t st) {
  if (!cl16_tr_applies(g)) return 0;
  if (bnr.part || scale_shift || res || relu) return 0;      // epilogues this kernel does not have: the tile kernel
  if (stat_sum && !cl16_tr_forward(g)) return 0;             // (the partial layout of such a launch is the tile kernel's)
  const int mt = g.Mrows / 16, kc = g.Cin_p / 32, pro = in_ss ? 1 : 0, epi = stat_sum ? 1 : 0;
  int rc = 0;
#define SLV_TR_K(MT_, KC_)                                                                               \
  do {                                                                                                   \
    if (pro == 0 && epi == 0) rc = tr_launch_one<MT_, KC_, 0, 0>(g, x, wl, y, in_ss, stat_sum, stat_sq, st); \
    else if (pro == 1 && epi == 0) rc = tr_launch_one<MT_, KC_, 1, 0>(g, x, wl, y, in_ss, stat_sum, stat_sq, st); \
    else if (pro == 0 && epi == 1) rc = tr_launch_one<MT_, KC_, 0, 1>(g, x, wl, y, in_ss, stat_sum, stat_sq, st); \
    else rc = tr_launch_one<MT_, KC_, 1, 1>(g, x, wl, y, in_ss, stat_sum, stat_sq, st);                    \
  } while (0)
  if (mt == 4 && kc == 5) SLV_TR_K(4, 5);
  else if (mt == 4 && kc == 2) SLV_TR_K(4, 2);
  else SLV_TR_K(9, 2);
#undef SLV_TR_K
  if (rc) return rc;
  rc = launch_check("slv_cl16_conv");
  return rc ? rc : 1;
}

}  // namespace slv
