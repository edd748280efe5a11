// 16-bit MFMA path: the WIDE layers (layers 2-4 of R(2+1)D-18, 128..1152 channels) -- forward and backward data of any
// kernel / stride / padding the lattice description of cl16.hpp expresses, as an implicit GEMM
// D[cout][pos] = sum_{tap, c} W[cout][tap][c] * act(X)[pos * stride + tap - pad][c]  in PING-PONG form.
//
// What bounded the tile kernels on these layers (csrc/conv_cl16.hip, conv_cl16_s3.hip: 450-950 TFLOP/s of 2 500): a stage
// is [issue loads | fragment reads | MFMAs | LDS writes | barrier] and the two workgroups of a CU run it in lockstep -- the
// phases of a stage follow each other, the matrix cores idle through all but one (profiles/r02_notes.md: no single resource
// binds).  Here ONE workgroup of 8 waves owns the CU and its two halves run the stage half a period apart:
//   * tile = 256 positions x BM = 2 x MTW x 16 output channels (256 or 288 or 128); wave (h, q) owns channel half h and
//     position quarter q: MTW x 4 accumulator tiles, MTW + 4 fragment reads for 4 MTW MFMAs per 32-channel chunk;
//   * group G0 = waves 0-3, G1 = waves 4-7: one wave of each group on every SIMD.  A wave alternates two slots,
//       R(c): write the activation pieces of chunk c + 1 to LDS, read the fragments of chunk c, wait for the requests in flight
//       M(c): the 4 MTW MFMAs of chunk c back to back (s_setprio 1), every operand already in registers
//     and every slot ends in s_barrier; G1 runs ONE barrier behind G0, so while G0's waves hold the matrix pipes G1's waves
//     do everything else, and vice versa.  The requests of later chunks -- the weights of chunk c + 2 by LDS-DMA
//     (buffer_load_dwordx4 ... lds, inline asm: 1 KiB per wave instruction), the activation pieces of chunk c + 3 into
//     registers -- and the BatchNorm + ReLU of the landed pieces sit where the measurements put them per form (below):
//     between the MFMAs of the M slot or at the end of the R slot behind its vmcnt wait.
//   * three chunk buffers in LDS ({weights [BM][64 B], activations [256][64 B]}, XOR-swizzled rows as in cl16.hpp: the
//     swizzle of a DMA'd image sits on the SOURCE address).  Hazards, with slot numbers (G0: R(c) = 2c, M(c) = 2c + 1; G1 one
//     later): chunk X's activations are written in R(X - 1) (slots 2X - 2 / 2X - 1), its weights are requested in R(X - 2) or
//     M(X - 2) (2X - 4 .. 2X - 2) and waited for at the end of R(X - 1); both are read from slot 2X on.  The buffer's previous
//     tenant, chunk X - 3, was last read in slot 2X - 5.  One s_waitcnt vmcnt(0) per R slot (LDS-DMA requests and register
//     loads do not complete in order with respect to each other -- profiles/r04_notes.md -- so the wait is never a counted one).
// Where it stands (profiles/r05_notes.md 4: ablations with tools/g8_ablate.py): the K loop runs at 0.94-0.97 PFLOP/s (the tile
// kernels: 0.5-0.86) but the epilogue -- 147 KB per workgroup through a transposed LDS tile, nothing on the CU to overlap it
// -- is a quarter of the launch at K = 1 152 and 40 % at K = 864: the kernel wins where the tile kernel ran (temporal, strided,
// pointwise convs with >= 256 output rows: layer 3 temporal forward 0.106 -> 0.063 ms at 64 clips) and loses to the patch kernel
// on the stride-1 spatial convs; cl16_g8_applies() dispatches accordingly.
// Epilogues: EPI 0 (affine / residual / ReLU -> bf16: eval-mode BatchNorm, backward data + addend) and EPI 1 (raw bf16 +
// BatchNorm statistics of the rounded tile on the matrix cores), per channel half through a transposed LDS tile, 16-byte
// stores along the channels.  Reference semantics: torchvision Conv3d / Conv2d forward and backward data as reached from
// /root/reference/model.py:93-100 (r2plus1d_18 layers 2-4) under main.py:151-153 (--use_fp16).
#include <type_traits>

#include "cl16.hpp"
#include "../../include/selavi_hip.h"

namespace slv {

constexpr int G8_BN = 256, G8_THREADS = 512, G8_NB = 3;
// Where the non-MFMA work of a chunk goes (tools/g8_ablate.py, profiles/r05_notes.md 4: every placement measured per shape):
//   weight DMA / activation requests: between the MFMAs of the M slot, or at the END of the R slot behind its vmcnt wait;
//   BatchNorm + ReLU of the landed pieces (PRO 1): at the LDS-write point in the R slot, or between the MFMAs (VALU in the
//   matrix pipe's shadow) -- the latter only where an M slot is long enough to cover ~70 VALU instructions (MTW >= 8).
// -1 = the per-instantiation choice below; 0 / 1 force one (ablation builds).
#ifndef SLV_G8_DMA_R_END
#define SLV_G8_DMA_R_END -1
#endif
#ifndef SLV_G8_LOADS_R_END
#define SLV_G8_LOADS_R_END -1
#endif
#ifndef SLV_G8_PRO_IN_M
#define SLV_G8_PRO_IN_M -1
#endif
#ifndef SLV_G8_OCC4
#define SLV_G8_OCC4 1          // the 128-row form (MTW = 4) without prologue at two workgroups per CU (128 registers): one's epilogue under
#endif                         // the other's loop (with the prologue the 128-register cap spills 24-32 bytes and buys 1.5 %)
#ifndef SLV_G8_DIRECT_EPI
#define SLV_G8_DIRECT_EPI 0    // 1: EPI 0 launches store their accumulators straight to memory (8-byte pieces, no LDS transposition, no barrier)
#endif
#ifndef SLV_G8_ABL
#define SLV_G8_ABL 0           // timing ablations (wrong results; tools/g8_ablate.sh): 1 no weight DMA, 2 no activation loads, 3 no LDS
#endif                         // writes, 4 no fragment reads, 5 no MFMAs, 6 no barrier behind the M slot, 7 no vmcnt wait, 8 no epilogue, 9 no K loop, 10 no output stores, 11 no LDS transposition writes, 12 no row-store phase

// one LDS-DMA piece: 64 lanes x 16 bytes, memory (voff + soff) -> LDS at lds_addr + 16 * lane
__device__ __forceinline__ void g8_dma16(unsigned lds_addr, unsigned voff, __amdgpu_buffer_rsrc_t rsrc, unsigned soff) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(lds_addr), "v"(voff), "s"(rsrc), "s"(soff)
               : "m0", "memory");
}
// (sched_barrier: the asm's "memory" clobber holds LDS / global accesses in place, not register-only instructions -- hipcc
//  moves MFMAs across an inline-asm wait or barrier otherwise, out of the slot they were written in)
__device__ __forceinline__ void g8_barrier() {
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
}
__device__ __forceinline__ void g8_wait_vm() {
  if (SLV_G8_ABL != 7) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

constexpr int g8_orow(int mtw) { return mtw * 32 + 16; }      // bytes per position row of the transposed half tile
constexpr size_t g8_lds_bytes(int mtw, int cin_p, bool pro) {
  const size_t loop = (size_t)G8_NB * (2 * mtw * 16 + G8_BN) * 64 + (pro ? (size_t)2 * cin_p * 4 : 0);
  const size_t epi = (size_t)G8_BN * g8_orow(mtw) + (size_t)(2 + 16) * mtw * 16 * 4 + G8_BN * 4;
  return loop > epi ? loop : epi;
}

template <int MTW, int PRO, int EPI>
__global__ __launch_bounds__(G8_THREADS, (SLV_G8_OCC4 && MTW == 4 && PRO == 0) ? 4 : 2) void conv_cl16_g8_kernel(const unsigned short* __restrict__ x,
                                                                      const unsigned short* __restrict__ wl,
                                                                      unsigned short* __restrict__ y,
                                                                      const float* __restrict__ in_ss,
                                                                      const float* __restrict__ scale_shift,
                                                                      const unsigned short* __restrict__ res, int relu,
                                                                      float* __restrict__ stat_sum,
                                                                      float* __restrict__ stat_sq, ClConv g, FastDiv dLw,
                                                                      FastDiv dLh, FastDiv dLt, FastDiv dGx) {
  constexpr int BMH = MTW * 16, BM = 2 * BMH;
  constexpr int ABYTES = BM * 64, STAGE = ABYTES + G8_BN * 64;
  // measured best per form (64 clips x 16 frames, ms of the train-mode forward / plain forward): layer 2.1 spatial 0.356 with the
  // prologue in the M slot and both requests at the end of R (0.396 with the first arrangement), layer 3.1 spatial 0.166 (0.191),
  // layer 3.1 temporal 0.063 (0.074); plain launches 0.391 -> 0.359 with the DMA at the end of R; the 128-row form keeps the
  // first arrangement (its M slot is 16 MFMAs: nothing to hide 70 VALU instructions behind: 0.197 vs 0.243)
  constexpr bool PRO_IN_M = SLV_G8_PRO_IN_M >= 0 ? SLV_G8_PRO_IN_M != 0 : (PRO == 1 && MTW >= 8);
  constexpr bool DMA_R_END = SLV_G8_DMA_R_END >= 0 ? SLV_G8_DMA_R_END != 0 : (PRO == 0 || MTW >= 8);
  constexpr bool LOADS_R_END = SLV_G8_LOADS_R_END >= 0 ? SLV_G8_LOADS_R_END != 0 : (PRO == 1 && MTW >= 8);
  constexpr int NDMA = BM / 16;                           // 1 KiB pieces of a chunk's weight image (16 rows each)
  constexpr int OROW = g8_orow(MTW);
  extern __shared__ __attribute__((aligned(1024))) unsigned char lds_raw[];
  typedef __attribute__((address_space(3))) void* lds_void;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2, wh = wave & 1, wq = (wave >> 1) & 1 | ((wave >> 2) << 1);      // group, channel half, position quarter
  const unsigned P = (unsigned)g.N * g.Lt * g.Lh * g.Lw;
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(
      (void*)x, 0, (int)((unsigned)g.N * g.Ti * g.Hi * g.Wi * g.Cin_p * 2u), 0x00020000);
  const __amdgpu_buffer_rsrc_t rwl = __builtin_amdgcn_make_buffer_rsrc((void*)wl, 0, (int)0x7FFFFFF0, 0x00020000);
  unsigned bx = blockIdx.x, by = blockIdx.y;
  if (XCD_REMAP) {        // every XCD gets a contiguous range of (channel block, position tile) units: the rows neighbouring
    const unsigned nb = gridDim.x * gridDim.y, lin = blockIdx.x + blockIdx.y * gridDim.x, q8 = nb >> 3, r8 = nb & 7,      // tiles re-read
                   xcd = lin & 7, loc = lin >> 3;                                                                         // for their shifted
    const unsigned unit = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + loc;                            // taps are in ITS L2
    by = fdiv(unit, dGx);
    bx = unit - by * gridDim.x;
  }
  const int m0 = by * BM;
  // ---- activation rows of this thread: 256 rows x 4 pieces of 16 B, 2 per thread (rows tid / 4 and tid / 4 + 128)
  int bt[2], bh[2], bw[2];
  unsigned bbase[2];
  const int piece = tid & 3, brow = tid >> 2;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const unsigned p = bx * G8_BN + brow + 128 * i;
    const unsigned q1 = fdiv(p, dLw), q2 = fdiv(q1, dLh), q = fdiv(q2, dLt);      // q = clip
    const int lw = p - q1 * g.Lw, lh = q1 - q2 * g.Lh, lt = q2 - q * g.Lt;
    bt[i] = lt * g.bmt + g.bot;
    bh[i] = lh * g.bmh + g.boh;
    bw[i] = lw * g.bmw + g.bow;
    bbase[i] = (((q * g.Ti + bt[i]) * g.Hi + bh[i]) * g.Wi + bw[i]) * (unsigned)(g.Cin_p * 2) + piece * 16u;
    if (p >= P) bt[i] = -(1 << 20);          // (bbase wraps for negative coordinates; only used when the tap is valid)
  }
  const int kcs = g.Cin_p >> 5, nch = g.ntaps * kcs;
  float* pro = (float*)(lds_raw + G8_NB * STAGE);                 // PRO 1: [2][Cin_p] scale, shift (zero beyond Cin)
  const unsigned lds_base = (unsigned)(unsigned long)(lds_void)lds_raw;
  // weight DMA: piece i = rows 16 i .. 16 i + 15 of the chunk's [BM][64 B] image; lane -> (row lane / 4, slot lane % 4), the
  // slot's swizzle on the source side (cl_swz(row) = (-(row >> 2)) & 3 does not depend on i)
  const unsigned voffA = (unsigned)((m0 + (lane >> 2)) * 64 + (((lane & 3) ^ ((-(lane >> 4)) & 3)) << 4));
  const unsigned bsw = (unsigned)((piece ^ ((-(tid >> 4)) & 3)) << 4);          // this thread's slot in its two activation rows
  const int fr = lane & 15, fk = lane >> 4, fsw = (fk ^ cl_swz(fr)) << 4;

  // K-step cursors: the chunk whose activation pieces are requested next, the chunk whose weights are requested next
  int tapL = 0, kcL = 0, cL = 0, tapD = 0, kcD = 0, cD = 0, bufD = 0;
  // the tap words of the two cursors, fetched in the R slots (scalar loads from the kernel arguments: with the fetch inside an M
  // slot the wave sat in s_waitcnt between its 4th and 5th MFMA for the whole scalar-memory round trip)
  int tpL = g.tap[0], tpD = g.tap[0];
  auto fetch_taps = [&]() __attribute__((always_inline)) {
    tpL = g.tap[tapL < g.ntaps ? tapL : 0];
    tpD = g.tap[tapD < g.ntaps ? tapD : 0];
  };
  u32x4 rb[2][2];
  unsigned okl = 0;                    // bit set * 2 + i: that piece lies inside the tensor
  int kcw[2] = {0, 0};                 // the channel chunk of the pieces in set s (prologue table index)
  // (the tap's part of a request -- validity of the two rows, offset of the shifted row -- is formed when the tap CHANGES, every
  //  Cin_p / 32 chunks, not per chunk: ~30 VALU instructions of every R slot otherwise)
  unsigned okT = 0, toffT = 0;
  auto issue_loads = [&](auto set_tag) __attribute__((always_inline)) {
    constexpr int S = decltype(set_tag)::value;
    if (cL < nch) {
      if (kcL == 0) {
        const int tp = tpL;
        const int dt = (tp & 15) - 8, dh = ((tp >> 4) & 15) - 8, dw = ((tp >> 8) & 15) - 8;
        toffT = (unsigned)(((dt * g.Hi + dh) * g.Wi + dw) * g.Cin_p * 2);
        okT = 0;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const bool ok = (unsigned)(bt[i] + dt) < (unsigned)g.Ti && (unsigned)(bh[i] + dh) < (unsigned)g.Hi &&
                          (unsigned)(bw[i] + dw) < (unsigned)g.Wi;
          okT |= (unsigned)ok << i;
        }
      }
      const unsigned toff = toffT + (unsigned)(kcL * 64);
      okl = (okl & ~(3u << (S * 2))) | (okT << (S * 2));
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        if (SLV_G8_ABL == 2) rb[S][i] = (u32x4){toff, bbase[i], 1u, 2u};
        else rb[S][i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, ((okT >> i) & 1u) ? bbase[i] + toff : 0xFFFFFFF0u, 0, 0));
      }
      kcw[S] = kcL;
      ++cL;
      if (++kcL == kcs) {
        kcL = 0;
        ++tapL;
      }
    }
  };
  auto issue_dma = [&]() __attribute__((always_inline)) {
    if (cD < nch) {
      const int slab = tpD >> 12;
      const unsigned soff = (unsigned)((slab * kcs + kcD) * g.Mrows) * 64u;
      const unsigned dst = lds_base + (unsigned)(bufD * STAGE);
#pragma unroll
      for (int t = 0; t < (NDMA + 7) / 8; ++t) {
        const int i = wave + 8 * t;
        if (i < NDMA && SLV_G8_ABL != 1)
          g8_dma16(__builtin_amdgcn_readfirstlane(dst + (unsigned)(i * 1024)), voffA, rwl,
                   __builtin_amdgcn_readfirstlane(soff + (unsigned)(i * 1024)));
      }
      ++cD;
      bufD = bufD == G8_NB - 1 ? 0 : bufD + 1;
      if (++kcD == kcs) {
        kcD = 0;
        ++tapD;
      }
    }
  };
  auto write_b = [&](auto set_tag, int buf) __attribute__((always_inline)) {
    constexpr int S = decltype(set_tag)::value;
    unsigned char* B = lds_raw + buf * STAGE + ABYTES;
    u32x4 v[2] = {rb[S][0], rb[S][1]};
    if constexpr (PRO == 1 && !PRO_IN_M) {                   // the producer's BatchNorm + ReLU, zero padding AFTER it
      float s[8], h[8];
      const float* sp = pro + kcw[S] * 32 + piece * 8;
      *(f32x4*)s = *(const f32x4*)sp;
      *(f32x4*)(s + 4) = *(const f32x4*)(sp + 4);
      *(f32x4*)h = *(const f32x4*)(sp + g.Cin_p);
      *(f32x4*)(h + 4) = *(const f32x4*)(sp + g.Cin_p + 4);
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const u32x4 t = affine_relu8(v[i], s, h);
        v[i] = ((okl >> (S * 2 + i)) & 1u) ? t : (u32x4){0u, 0u, 0u, 0u};
      }
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) *(u32x4*)(B + (brow + 128 * i) * 64 + bsw) = v[i];
  };

  // PRO 1, PRO_IN_M: the pieces of set S (landed) -> relu(x s + h) in place, zero where the tap left the tensor
  auto transform = [&](auto set_tag) __attribute__((always_inline)) {
    constexpr int S = decltype(set_tag)::value;
    if constexpr (PRO == 1) {
      float sc[8], sh[8];
      const float* sp = pro + kcw[S] * 32 + piece * 8;
      *(f32x4*)sc = *(const f32x4*)sp;
      *(f32x4*)(sc + 4) = *(const f32x4*)(sp + 4);
      *(f32x4*)sh = *(const f32x4*)(sp + g.Cin_p);
      *(f32x4*)(sh + 4) = *(const f32x4*)(sp + g.Cin_p + 4);
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const u32x4 t = affine_relu8(rb[S][i], sc, sh);
        rb[S][i] = ((okl >> (S * 2 + i)) & 1u) ? t : (u32x4){0u, 0u, 0u, 0u};
      }
    }
  };

  f32x4 acc[MTW][4];
#pragma unroll
  for (int i = 0; i < MTW; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  constexpr std::integral_constant<int, 0> S0{};
  constexpr std::integral_constant<int, 1> S1{};
  // ---- prologue of the pipeline: chunks 0 and 1 requested, chunk 0's activations in LDS, chunk 2's pieces in flight
  issue_dma();
  fetch_taps();
  issue_dma();
  issue_loads(S0);
  fetch_taps();
  issue_loads(S1);
  fetch_taps();
  if constexpr (PRO == 1) {
    for (int i = tid; i < 2 * g.Cin_p; i += G8_THREADS) {
      const int c = i % g.Cin_p, which = i / g.Cin_p;
      pro[i] = c < g.Cin ? in_ss[which * g.Cin + c] : 0.f;
    }
    g8_barrier();
  }
  g8_wait_vm();
  if (PRO_IN_M) {
    transform(S0);
    transform(S1);
  }
  write_b(S0, 0);
  issue_loads(S0);
  fetch_taps();
  g8_barrier();
  if (grp == 1 && SLV_G8_ABL != 6) g8_barrier();    // G1 runs one slot behind G0 from here on

  int bufR = 0;                                     // buffer of the chunk being multiplied; chunk c + 1 lives in the next one
  auto iter = [&](auto set_tag) __attribute__((always_inline)) {      // set_tag: parity of chunk c + 1 (= of chunk c + 3)
    constexpr int S = decltype(set_tag)::value;
    const int bufW = bufR == G8_NB - 1 ? 0 : bufR + 1;
    // ---- R(c): the pieces of chunk c + 1 (requested two chunks ago) go to LDS, the fragments of chunk c come out, then the
    // wait for everything requested a period ago and -- in the forms that have them here -- the new requests.  (All requests at
    // the TOP of the slot measured worse, 0.42 against 0.36 ms on layer 2.1: the slot's own work waits behind the DMA issue.)
    const unsigned char* A = lds_raw + bufR * STAGE + (wh * BMH + fr) * 64 + fsw;
    const unsigned char* B = lds_raw + bufR * STAGE + ABYTES + (wq * 64 + fr) * 64 + fsw;
    bf16x8 a[MTW], b[4];
    u32x4 keep[2] = {rb[S][0], rb[S][1]};           // chunk c + 1's pieces leave the request registers
    const unsigned keep_ok = (okl >> (S * 2)) & 3u;
    const int keep_kc = kcw[S];
    fetch_taps();
    {
      unsigned char* Bw = lds_raw + bufW * STAGE + ABYTES;
      if constexpr (PRO == 1 && !PRO_IN_M) {                   // the producer's BatchNorm + ReLU, zero padding AFTER it
        float sc[8], sh[8];
        const float* sp = pro + keep_kc * 32 + piece * 8;
        *(f32x4*)sc = *(const f32x4*)sp;
        *(f32x4*)(sc + 4) = *(const f32x4*)(sp + 4);
        *(f32x4*)sh = *(const f32x4*)(sp + g.Cin_p);
        *(f32x4*)(sh + 4) = *(const f32x4*)(sp + g.Cin_p + 4);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const u32x4 t = affine_relu8(keep[i], sc, sh);
          keep[i] = ((keep_ok >> i) & 1u) ? t : (u32x4){0u, 0u, 0u, 0u};
        }
      }
      if (SLV_G8_ABL != 3) {
#pragma unroll
        for (int i = 0; i < 2; ++i) *(u32x4*)(Bw + (brow + 128 * i) * 64 + bsw) = keep[i];
      } else {
        asm volatile("" ::"v"(keep[0]), "v"(keep[1]));
      }
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < 4; ++j) b[j] = SLV_G8_ABL == 4 ? __builtin_bit_cast(bf16x8, (u32x4){(unsigned)(size_t)B, (unsigned)j, 1u, 2u}) : *(const bf16x8*)(B + j * 1024);
#pragma unroll
    for (int i = 0; i < MTW; ++i) a[i] = SLV_G8_ABL == 4 ? __builtin_bit_cast(bf16x8, (u32x4){(unsigned)(size_t)A, (unsigned)i, 3u, 4u}) : *(const bf16x8*)(A + i * 1024);
    {
      g8_wait_vm();                                 // the requests of M(c - 1) / the end of R(c - 1): chunk c + 1's weights, chunk c + 2's pieces
      if (DMA_R_END) issue_dma();
      if (LOADS_R_END) issue_loads(set_tag);
    }
    g8_barrier();
    // ---- M(c): the MFMAs of chunk c (and whatever of the requests / the prologue this form places between them)
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int i = 0; i < MTW; ++i) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (SLV_G8_ABL == 5) acc[i][j][0] += __builtin_bit_cast(float, __builtin_bit_cast(u32x4, a[i])[0] ^ __builtin_bit_cast(u32x4, b[j])[0]);
        else acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
      }
      if (!DMA_R_END && i == 0) issue_dma();                // chunk c + 2's weights (its buffer was last read in R(c - 1))
      if (!LOADS_R_END && i == 1) issue_loads(set_tag);     // chunk c + 3's pieces (the set R(c) just emptied)
      if (PRO_IN_M && i == 2) transform(std::integral_constant<int, S ^ 1>{});   // chunk c + 2's pieces (landed: waited for in R(c))
    }
    __builtin_amdgcn_s_setprio(0);
    if (SLV_G8_ABL != 6) g8_barrier();
    bufR = bufW;
  };
  for (int c = 0; c < (SLV_G8_ABL == 9 ? 0 : nch); c += 2) {
    iter(S1);
    if (c + 1 < nch) iter(S0);
  }
  if (grp == 0 && SLV_G8_ABL != 6) g8_barrier();
  g8_wait_vm();
  g8_barrier();

  if (SLV_G8_ABL == 8) {                            // (every accumulator stays alive: an ablation must not delete the MFMAs)
#pragma unroll
    for (int i = 0; i < MTW; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) asm volatile("" ::"v"(acc[i][j]));
    return;
  }
  if constexpr (EPI == 0 && SLV_G8_DIRECT_EPI) {
    // ---- direct epilogue: every wave stores ITS tiles as they are -- a lane holds 4 consecutive channels of one position
    // (8 bytes), the 4 k-groups of a position 32 contiguous bytes -- no LDS, no barrier: with one workgroup per CU nothing
    // overlaps an epilogue, so its phases (transpose, barrier, row stores) cost their full latency each
    unsigned opj[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const unsigned p = bx * G8_BN + wq * 64 + j * 16 + fr;
      unsigned o = 0xFFFFFFFFu;
      if (p < P) {
        const unsigned q1 = fdiv(p, dLw), q2 = fdiv(q1, dLh), q = fdiv(q2, dLt);
        const int lw = p - q1 * g.Lw, lh = q1 - q2 * g.Lh, lt = q2 - q * g.Lt;
        o = ((q * g.To + lt * g.omt + g.oot) * g.Ho + lh * g.omh + g.ooh) * g.Wo + lw * g.omw + g.oow;
      }
      opj[j] = o;
    }
    const int c0 = m0 + wh * BMH;
#pragma unroll
    for (int i = 0; i < MTW; ++i) {
      const int co = c0 + i * 16 + fk * 4;
      f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
      if (scale_shift) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          sc[r] = co + r < g.Cout ? scale_shift[co + r] : 0.f;
          sh[r] = co + r < g.Cout ? scale_shift[g.Cout + co + r] : 0.f;
        }
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const unsigned op = opj[j];
        if (op == 0xFFFFFFFFu || co >= g.Cout_p) continue;
        uint2 rr = make_uint2(0u, 0u);
        if (res) rr = *(const uint2*)(res + (size_t)op * g.Cout_p + co);
        float v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float t = acc[i][j][r];
          if (scale_shift) t = __builtin_fmaf(t, sc[r], sh[r]);
          if (res) t += bf2f((unsigned short)((r < 2 ? rr.x : rr.y) >> ((r & 1) * 16)));
          if (relu) t = fmaxf(t, 0.f);
          v[r] = t;
        }
        *(uint2*)(y + (size_t)op * g.Cout_p + co) = make_uint2(pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]));
      }
    }
    if (by == gridDim.y - 1 && g.Mrows < g.Cout_p && wh == 1) {      // padding channels no tile covers: zeros
      for (int c = g.Mrows + fk * 4; c < g.Cout_p; c += 16)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (opj[j] != 0xFFFFFFFFu) *(uint2*)(y + (size_t)opj[j] * g.Cout_p + c) = make_uint2(0u, 0u);
    }
    return;
  }
  // ---- epilogue, one channel half at a time: the half's tiles transposed through LDS ([position][cout] rows), statistics
  // of the rounded tile on the matrix cores (8 waves x 32 rows), 16-byte stores along a position's channels
  unsigned char* ot = lds_raw;                                    // [256][OROW]
  float* ssl = (float*)(lds_raw + G8_BN * OROW);                  // EPI 0: [2][BMH] scale, shift of the half
  float* red = ssl + 2 * BMH;                                     // EPI 1: [8 waves][2][BMH] partials
  unsigned* opos = (unsigned*)(red + 16 * BMH);                   // [256] output position (row index) or ~0
  if (tid < G8_BN) {
    const unsigned p = bx * G8_BN + tid;
    unsigned o = 0xFFFFFFFFu;
    if (p < P) {
      const unsigned q1 = fdiv(p, dLw), q2 = fdiv(q1, dLh), q = fdiv(q2, dLt);
      const int lw = p - q1 * g.Lw, lh = q1 - q2 * g.Lh, lt = q2 - q * g.Lt;
      o = ((q * g.To + lt * g.omt + g.oot) * g.Ho + lh * g.omh + g.ooh) * g.Wo + lw * g.omw + g.oow;
    }
    opos[tid] = o;
  }
#pragma unroll 1
  for (int hh = 0; hh < 2; ++hh) {
    const int c0 = m0 + hh * BMH;                                 // first channel of the half
    if (EPI == 0 && scale_shift) {
      for (int i = tid; i < 2 * BMH; i += G8_THREADS) {
        const int c = c0 + (i % BMH);
        ssl[i] = c < g.Cout ? scale_shift[(i / BMH) * g.Cout + c] : 0.f;
      }
    }
    g8_barrier();                                                 // opos / ssl written; the previous half's rows stored
    if (wh == hh) {
#pragma unroll
      for (int i = 0; i < MTW; ++i) {
        const int co = c0 + i * 16 + fk * 4;
        f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
        if (EPI == 0 && scale_shift) {
          sc = *(const f32x4*)(ssl + i * 16 + fk * 4);
          sh = *(const f32x4*)(ssl + BMH + i * 16 + fk * 4);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int pl = wq * 64 + j * 16 + fr;                   // position inside the block
          float v[4];
          if constexpr (EPI == 0) {
            const unsigned op = opos[pl];
            uint2 rr = make_uint2(0u, 0u);
            if (res && op != 0xFFFFFFFFu && co < g.Cout_p) rr = *(const uint2*)(res + (size_t)op * g.Cout_p + co);
            if (!scale_shift && !relu) {      // backward data: the accumulator (+ addend) as it is
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                v[r] = acc[i][j][r];
                if (res) v[r] += bf2f((unsigned short)((r < 2 ? rr.x : rr.y) >> ((r & 1) * 16)));
              }
            } else {
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                float t = __builtin_fmaf(acc[i][j][r], sc[r], sh[r]);
                if (res) t += bf2f((unsigned short)((r < 2 ? rr.x : rr.y) >> ((r & 1) * 16)));
                if (relu) t = fmaxf(t, 0.f);
                v[r] = t;
              }
            }
          } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = acc[i][j][r];      // rows >= Cout: zero weights -> 0
          }
          if (SLV_G8_ABL == 11) asm volatile("" ::"v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]));
          else *(uint2*)(ot + pl * OROW + (i * 16 + fk * 4) * 2) = make_uint2(pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]));
        }
      }
    }
    g8_barrier();
    if constexpr (EPI == 1) {
      wave_tile_stats<MTW>(ot + wave * 32 * OROW, OROW, lane, red + (wave * 2 + 0) * BMH, red + (wave * 2 + 1) * BMH);
      g8_barrier();
      for (int i = tid; i < 2 * BMH; i += G8_THREADS) {
        const int c = i % BMH, which = i / BMH;
        if (c0 + c < g.Cout) {
          float t = red[which * BMH + c];
#pragma unroll
          for (int w = 1; w < 8; ++w) t += red[(w * 2 + which) * BMH + c];      // the 8 waves' partials in fixed order
          (which ? stat_sq : stat_sum)[(size_t)(c0 + c) * gridDim.x + bx] = t;   // [Cout][tiles]
        }
      }
    }
    // rows of this half in the output: channels [c0, c0 + BMH) clipped to Cout_p; the last half of the last channel block
    // also zero-fills the padding channels no tile covers (Mrows < Cout_p)
    const int c_hi = min(c0 + BMH, g.Cout_p);
    const int c_end = (by == gridDim.y - 1 && hh == 1) ? g.Cout_p : c_hi;
    const int pieces = (c_end - c0) >> 3;                 // 16-byte pieces per position
    if (pieces > 0 && SLV_G8_ABL != 12) {
      const int rpp = G8_THREADS / pieces, pl0 = tid / pieces, pc = tid - pl0 * pieces;
      if (pl0 < rpp) {
        const bool cval = c0 + pc * 8 < c_hi;
        const unsigned char* src = ot + pl0 * OROW + pc * 16;
        unsigned char* yb = (unsigned char*)y;
        const unsigned coff = (unsigned)(c0 + pc * 8) * 2u, rowb = (unsigned)g.Cout_p * 2u;
        for (int pl = pl0; pl < G8_BN; pl += rpp, src += rpp * OROW) {
          const unsigned op = opos[pl];
          if (op == 0xFFFFFFFFu) continue;
          u32x4 val = {0u, 0u, 0u, 0u};
          if (cval) val = *(const u32x4*)src;
          if (SLV_G8_ABL == 10) { if (val[0] == 0x12345678u) *(u32x4*)(yb + op * rowb + coff) = val; }
          else *(u32x4*)(yb + op * rowb + coff) = val;
        }
      }
    }
  }
}

// MTW (16-row tiles per channel HALF of a block) from the weight layout's rows: blocks of 288, 256 or 128 rows
static int g8_mtw(const ClConv& g) {
  if (g.Mrows % 288 == 0) return 9;
  if (g.Mrows % 256 == 0) return 8;
  if (g.Mrows % 128 == 0) return 4;
  return 0;
}
// SELAVI_CL16_G8: 0 = off, 1 (default) = launches that fill the chip, force = every launch the kernel can express (tests);
// slv_cl16_g8_mode() changes it at run time
static int g8_mode_value = -1;
static int g8_mode() {
  if (g8_mode_value < 0) {
    const char* e = getenv("SELAVI_CL16_G8");
    g8_mode_value = !e ? 1 : (e[0] == '0' ? 0 : ((e[0] == 'f' || e[0] == 'F' || e[0] == '2') ? 2 : 1));
  }
  return g8_mode_value;
}
int cl16_g8_set_mode(int mode) {
  const int prev = g8_mode();
  if (mode >= 0 && mode <= 2) g8_mode_value = mode;
  return prev;
}
bool cl16_g8_applies(const ClConv& g) {
  const int mode = g8_mode();
  if (!mode) return false;
  const int mtw = g8_mtw(g);
  if (!mtw || g.ntaps < 1 || g.Cin_p < 32 || g.Cin_p > CL_PRO_MAXC) return false;
  if (g8_lds_bytes(mtw, g.Cin_p, true) > 160 * 1024) return false;
  const long long P = (long long)g.N * g.Lt * g.Lh * g.Lw;
  const long long blocks = ((P + G8_BN - 1) / G8_BN) * (g.Mrows / (32 * mtw));
  if (mode == 1) {
    // What the A/B of the layer shapes says (tools/g8_ab.py, 64 clips x 16 frames and cfg5's 128 x 32; profiles/r05_notes.md 4):
    // this kernel wins the FORWARD launches the tile kernel (csrc/conv_cl16.hip) had -- temporal, strided and pointwise convs
    // with >= 256 output rows: layer 3 temporal 0.106 -> 0.063 ms, layer 4.0 spatial 0.136 -> 0.085 -- and loses to the patch
    // kernel on the stride-1 spatial convs (activations read once for nine taps there), on the 128-row launches and on the
    // backward data.  One workgroup per CU: a launch of fewer than ~a round of 256 leaves CUs idle that the 128-position
    // tiles would fill; narrow contractions (the stems) have nothing for the pipeline to overlap.
    if (mtw < 8 || blocks < 160 || g.Cin_p * g.ntaps < 256) return false;
    if (g.flags & 1) {
      if (cl16_s3_applies(g)) return false;
    } else {
      // backward data: only on the 7 x 7 maps of layer 4, where the patch kernel's 128-position tiles are mostly halo and the
      // tile kernel's launches are short (128 clips x 32 frames: layer 4.1 spatial 0.307 -> 0.239 ms, layer 4.0 strided spatial
      // 0.449 -> 0.273; the layer-3 launches lose 10-20 %: profiles/r05_g8_ab_cfg5_l3_l4.txt)
      if (g.Hi * g.Wi > 64) return false;
    }
  }
  return true;
}
int cl16_g8_positions() { return G8_BN; }

template <int MTW, int PRO, int EPI>
static int g8_launch_one(const ClConv& g, const void* x, const void* wl, void* y, const float* in_ss,
                         const float* scale_shift, const void* res, int relu, float* stat_sum, float* stat_sq,
                         hipStream_t st) {
  const size_t lds = g8_lds_bytes(MTW, g.Cin_p, PRO != 0);
  static bool attr_set = false;                          // per instantiation; idempotent, so a race is harmless
  if (!attr_set) {
    SLV_HIP(hipFuncSetAttribute((const void*)conv_cl16_g8_kernel<MTW, PRO, EPI>,
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_set = true;
  }
  const unsigned P = (unsigned)g.N * g.Lt * g.Lh * g.Lw;
  dim3 grid((P + G8_BN - 1) / G8_BN, g.Mrows / (32 * MTW));
  hipLaunchKernelGGL((conv_cl16_g8_kernel<MTW, PRO, EPI>), grid, dim3(G8_THREADS), lds, st, (const unsigned short*)x,
                     (const unsigned short*)wl, (unsigned short*)y, in_ss, scale_shift, (const unsigned short*)res, relu,
                     stat_sum, stat_sq, g, make_fastdiv(g.Lw), make_fastdiv(g.Lh), make_fastdiv(g.Lt), make_fastdiv(grid.x));
  return 0;
}

// returns 1 when the launch was taken, 0 when it does not apply, < 0 on error
int cl16_g8_try(const ClConv& g, const void* x, const void* wl, void* y, const float* in_ss, const float* scale_shift,
                const void* res, int relu, float* stat_sum, float* stat_sq, const ClBnr& bnr, hipStream_t st) {
  if (!cl16_g8_applies(g)) return 0;
  if (bnr.part)        // (slv_cl16_conv_nblk reports THIS kernel's tile count for the geometry: no silent change of kernel)
    return fail(-2, "%s: the 8-wave conv kernel has no fused BatchNorm-backward sums (SELAVI_CL16_FUSE_BNR=1 needs SELAVI_CL16_G8=0)",
                "slv_cl16_conv");
  const int mtw = g8_mtw(g), pro = in_ss ? 1 : 0, epi = stat_sum ? 1 : 0;
  int rc = 0;
#define SLV_G8_K(MTW_)                                                                                                    \
  do {                                                                                                                    \
    if (pro == 0 && epi == 0) rc = g8_launch_one<MTW_, 0, 0>(g, x, wl, y, in_ss, scale_shift, res, relu, stat_sum, stat_sq, st); \
    else if (pro == 1 && epi == 0) rc = g8_launch_one<MTW_, 1, 0>(g, x, wl, y, in_ss, scale_shift, res, relu, stat_sum, stat_sq, st); \
    else if (pro == 0 && epi == 1) rc = g8_launch_one<MTW_, 0, 1>(g, x, wl, y, in_ss, scale_shift, res, relu, stat_sum, stat_sq, st); \
    else rc = g8_launch_one<MTW_, 1, 1>(g, x, wl, y, in_ss, scale_shift, res, relu, stat_sum, stat_sq, st);                \
  } while (0)
  if (mtw == 9) SLV_G8_K(9);
  else if (mtw == 8) SLV_G8_K(8);
  else SLV_G8_K(4);
#undef SLV_G8_K
  if (rc) return rc;
  rc = launch_check("slv_cl16_conv");
  return rc ? rc : 1;
}

}  // namespace slv
