// fp32 implicit-GEMM convolution on the bf16 matrix cores with SPLIT OPERANDS ("x3"): every fp32 operand element is cut
// into three bf16 pieces  x = x1 + x2 + x3  (8 + 8 + 8 significand bits: exact), and a product  a*b  is evaluated as the
// six partial products  a1 b1 + a1 b2 + a2 b1 + a1 b3 + a2 b2 + a3 b1  on v_mfma_f32_16x16x32_bf16 with fp32 accumulation.
// A product of two bf16 is exact in fp32, so the only terms dropped are a2 b3 + a3 b2 + a3 b3 <= 3 * 2^-24 |a b| -- the size
// of ONE fp32 rounding of the product; the accumulation is fp32 exactly as in the native fp32 MFMA (igemm.hpp).  Measured
// against fp64 the kernel sits where the native kernel sits (tests/test_ops_gpu.py holds both to the same 5e-6).
//
// Why: gfx950 has no TF32/xf32 path and its fp32-input MFMA runs at the fp32 VECTOR rate (157 TF, 1/16 of the bf16 rate);
// six bf16 MFMAs of K = 32 take 6 x 16 = 96 cycles where the eight 16x16x4_f32 MFMAs of the same 32 k take 256:
// the fp32 convs of the SeLaVi step (98 % of its FLOPs, all MFMA-bound: SURVEY.md 8d) get a 2.67x higher ceiling
// (2.5 PF / 6 = 417 "fp32-equivalent" TFLOP/s).  Operands stay the reference's fp32 N,C,T,H,W tensors in HBM; the split
// happens in registers on the way into LDS (after the BatchNorm + ReLU prologue), weights included.
//
// Same launch geometry, tables, prologue and epilogues as igemm.hpp's MODE_CONV with tap-major K (KORD_TAP):
//   forward conv and backward data of every layer but the two stems (those keep the native fp32 kernel).
// Differences: a K chunk is 32 deep = TWO consecutive 16-channel groups of the tap-major order (each with its own tap:
// the table keeps one entry per 16-deep group); LDS holds three bf16 planes per operand in [row][32 k] rows of 64 bytes,
// XOR-swizzled (cl_swz) so that the ds_read_b128 fragment reads are conflict-free; one fragment read feeds an MFMA of
// K = 32.  Reference semantics: torchvision Conv3d/Conv2d forward and backward data as reached from
// /root/reference/model.py:95,114 and main.py:284-301.
#pragma once
#include "cl16.hpp"
#include "igemm.hpp"

namespace slv {

// x (fp32 bits) -> three bf16 pieces by truncation (each piece takes the next 8 significand bits: x1 + x2 + x3 == x exactly)
__device__ __forceinline__ void split3(float x, unsigned& h, unsigned& m, unsigned& l) {
  const unsigned u = __float_as_uint(x);
  h = u & 0xffff0000u;
  const float r1 = x - __uint_as_float(h);            // exact
  const unsigned u1 = __float_as_uint(r1);
  m = u1 & 0xffff0000u;
  const float r2 = r1 - __uint_as_float(m);           // exact, <= 8 significant bits: a bf16
  l = __float_as_uint(r2);
}
// "x2" (the eval-mode feature pass, opt-in; NP = 2 below): two pieces instead of three.  h = the upper 8 significand bits by
// truncation, m = the remainder ROUNDED to bf16: x = h + m (1 + e), |e| <= 2^-9 on a term <= 2^-8 |x| -> |x - h - m| <= 2^-17 |x|;
// a product then takes the three partial products a1 b1 + a1 b2 + a2 b1 (dropped: a2 b2 <= 2^-16 |a b| and the weights' third
// piece, <= 2^-16): 16-17 significand bits -- 30 x finer than TF32 -- at HALF the matrix-core work of the exact split.  Never
// used by the training step.  Two elements at a time: the packed dwords of the h and m planes.
__device__ __forceinline__ void split2_pair(float x0, float x1, unsigned& hpk, unsigned& mpk) {
  const unsigned u0 = __float_as_uint(x0) & 0xffff0000u, u1 = __float_as_uint(x1) & 0xffff0000u;
  hpk = __builtin_amdgcn_perm(u1, u0, 0x07060302u);
  mpk = pack_bf2(x0 - __uint_as_float(u0), x1 - __uint_as_float(u1));       // exact differences, one RN each
}
// One LDS-DMA piece: 64 lanes x 16 bytes, memory (buffer offset voff + soff) -> LDS at lds_addr + 16 * lane.  Inline asm: through
// the builtin, hipcc orders every LDS read it can see behind a DMA "that may alias" with s_waitcnt vmcnt(0) -- the whole
// round trip in front of the chunk's MFMAs.  The compiler does not count these requests: the consumer waits explicitly.
__device__ __forceinline__ void x3_dma16(unsigned lds_addr, unsigned voff, __amdgpu_buffer_rsrc_t rsrc, unsigned soff) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(lds_addr), "v"(voff), "s"(rsrc), "s"(soff)
               : "m0");      // m0 is written (declared: the compiler may keep a value of its own there -- readlane / movrel / its own
                             //  LDS-DMA builtins); no "memory" clobber: it would turn every uniform load behind it -- the tap
                             //  table -- into a vector load + vmcnt(0).  What orders the LDS reads of a stage behind its DMAs is the
                             //  explicit s_waitcnt vmcnt + barrier of the chunk loop, and tools/x3_asm_check.py checks the ISA for it
}
// A slot boundary of the ping-pong form (8-wave workgroups, below): LDS traffic drained, s_barrier; the sched_barriers keep
// register-only instructions (MFMAs) on their side of it (an asm's "memory" clobber holds memory accesses only).
__device__ __forceinline__ void x3_pp_barrier() {
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
}
// upper halves of two dwords -> one dword (lo = a's bf16, hi = b's bf16)
__device__ __forceinline__ unsigned pack_hi16(unsigned a, unsigned b) { return __builtin_amdgcn_perm(b, a, 0x07060302u); }

// MT x NT tiles of 16 x 16 per wave, block (MT*16) x (NT*64), 256 threads = 4 waves; wave w owns ALL MT row tiles of ITS
// NT*16 columns.  What the timeline of the first version said (tools/x3_trace.py: a K chunk took 15 000 cycles for 1 728 of
// MFMA work per wave): issuing the operand traffic, two barriers and an exposed LDS-DMA round trip per chunk -- so:
// A (weights): the per-step x3 image (conv_common.hpp: x3_image_floats -- three bf16 planes, already in the swizzled LDS row
//   layout) goes memory -> LDS by LDS-DMA, 1 KiB per wave instruction (no registers, no VALU, no ds_write), DOUBLE buffered:
//   the pieces of chunk c + 1 are requested before the MFMAs of chunk c.  ONE barrier per chunk.
// B (activations): never touches LDS.  A wave's columns are its own, so every lane loads exactly the elements of ITS MFMA
//   B fragments (column lane % 16 of each of the NT column tiles, the 8 channels of k group lane / 16: fp32 N,C,T,H,W,
//   64-byte runs per channel row), applies BatchNorm + ReLU, cuts them into the three bf16 pieces in registers (8.5 VALU
//   instructions per element, amortised over the MT row tiles) and those registers ARE the operands of the next chunk's
//   MFMAs: no ds_write, no ds_read, no barrier for B.
// The BatchNorm scale / shift table of the gathered tensor sits in LDS (the first version fetched 32 scalars per chunk with
// ~300 SALU instructions of address arithmetic in front of the MFMAs).
#ifdef SLV_X3_TRACE      // timeline of wave 0 of the first 64 workgroups (tools/x3_trace.py): s_memtime stamps
__device__ unsigned long long slv_x3_trace_buf[64][160];
#define X3_T(slot) do { if (trace) trace[slot] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define X3_T(slot)
#endif
#ifndef SLV_X3_ABL
#define SLV_X3_ABL 0      // timing ablations (wrong results): 1 no B loads, 2 no A DMA, 3 no split, 4 no MFMA
#endif
constexpr int X3_MAXC = 1152;     // widest gathered tensor of the two trunks (prologue table in LDS)
// WAVES: 4 or 8 waves per workgroup (8: the A stage is fetched once per (NT*128) columns instead of per (NT*64) -- the
// weight traffic L2 -> LDS per MFMA halves; same LDS, same waves per SIMD as two 4-wave workgroups)
// FUSE: B two chunks deep with the split between the MFMAs (above); false: one chunk deep, the split behind the chunk's
// MFMAs -- ~90 registers fewer (no second operand set, none of the rematerialisation the long live ranges cause: 140
// instead of 290 VALU instructions per chunk), which buys a third workgroup per CU for the tiles whose LDS admits it (MT <= 8).
// PING-PONG (WAVES == 8 with the lean pipeline; an experiment kept behind -DSLV_X3_PP=1, see launch_igemm3): what the counters said of the 4-wave forms -- MFMA busy 50 %, two waves per
// SIMD that run in phase, each stalled while the other holds the pipe it wants (profiles/r04_notes.md 2) -- is a scheduling
// problem between the two waves of a SIMD, not a resource limit.  One 8-wave workgroup owns the CU; an iteration is cut into
// an M slot (the A-stage DMA and the raw B loads of chunk c + 1 requested, then the 6 MT NT MFMAs of chunk c back to back at
// s_setprio 1) and an S slot (chunk c + 1's raw elements: BatchNorm + ReLU, cut into the three bf16 planes), each ending in
// s_barrier, and waves 4-7 run ONE barrier behind waves 0-3: a SIMD's two waves are always in opposite slots, its matrix
// pipe sees M slots back to back while the VALU does the other wave's split.  Hazards (slots: waves 0-3 M(c) = 2c, S(c) = 2c + 1,
// waves 4-7 one later): stage c & 1 is read in slots 2c and 2c + 1; the DMA pieces of chunk c + 1 go into the other stage
// from slot 2c on (its tenant, chunk c - 1, was last read in slot 2c - 1) and are waited for by their issuing wave at the end
// of its M(c) (slots 2c / 2c + 1): landed before slot 2c + 2.
// NP: bf16 pieces per operand -- 3 (exact split, 6 partial products: every training launch) or 2 (3 partial products: the
// opt-in eval-mode feature pass, see split2_pair).  EPI_EVAL (eval-mode forward with BatchNorm folded into the weights, selavi_amd/
// infer32.py): y = acc + bias[row] (+ E) and ReLU if g.epi_relu -- conv + BN (+ residual) + ReLU of a torchvision block in
// one launch, no statistics, no prologue on the consumer's side.
template <int MT, int NT, int PRO, int EPI, int WAVES, int OCC, bool FUSE, int NP = 3>
__global__ __launch_bounds__(64 * WAVES, OCC) void igemm3_kernel(const IgemmArgs g) {
  static_assert(NP == 2 || NP == 3, "two or three bf16 pieces per operand");
  constexpr int BM = MT * 16, BN = NT * 16 * WAVES, NTHR = 64 * WAVES;
  constexpr bool PP = WAVES == 8 && !FUSE;
  constexpr int NPROD = NP == 3 ? 6 : 3;
  constexpr int A_BYTES = NP * BM * 64;
  constexpr int EPI_BYTES = (2 * WAVES + 4) * BM * 4;
#ifndef SLV_X3_LDS_PAD
#define SLV_X3_LDS_PAD 0          // experiment: extra LDS bytes per workgroup (forces fewer workgroups per CU)
#endif
  constexpr int SMEM = (2 * A_BYTES > EPI_BYTES ? 2 * A_BYTES : EPI_BYTES) + SLV_X3_LDS_PAD;
  constexpr int NDMA = NP * MT;         // 1 KiB pieces of an A stage (16 rows of one plane each; NP = 2: the image's first two planes)
  __shared__ __attribute__((aligned(1024))) unsigned char smem[SMEM];
  // dynamic LDS: the tap table of this launch ([Kd / 16 + 4] entries {offset, tap | first channel << 8}: read per chunk
  // without a vector-memory round trip) and, PRO_ACT, [2][CbP] scale, shift (padding channels 0)
  extern __shared__ __attribute__((aligned(16))) int dyn_lds[];
  typedef __attribute__((address_space(3))) void* lds_void;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

  int mblk, nblk, split;
  {
    const int nb = gridDim.x, id = blockIdx.x;
    const int q = nb >> 3, r = nb & 7, xcd = id & 7, loc = id >> 3;
    const int nid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
    const int per = g.nblkM * g.nblkN;
    split = nid / per;
    const int rem_ = nid - split * per;
    mblk = rem_ % g.nblkM;
    nblk = rem_ / g.nblkM;
  }
  const int m0 = mblk * BM;
  const long long n0 = (long long)nblk * BN;
  const int mrem = g.M - m0;
  const int Mp = (g.M + 15) & ~15;
#ifdef SLV_X3_TRACE
  unsigned long long* trace = (tid == 0 && blockIdx.x < 64 && PRO == PRO_ACT) ? slv_x3_trace_buf[blockIdx.x] : nullptr;
  X3_T(0);
#endif

  const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc((void*)g.A, 0, (int)g.A_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rB = __builtin_amdgcn_make_buffer_rsrc((void*)g.B, 0, (int)g.B_bytes, 0x00020000);

  const int fi = lane & 15, fk = lane >> 4;     // fragment lane: row / column fi, k group fk (8 k each)
  const int sc = fk >> 1;                       // which of the chunk's two 16-channel groups this lane's k group is in
  const int cof = (fk & 1) * 8;                 // first channel inside it
  const int CbP = (g.Cb + 15) & ~15;
  const int ntab = ((g.Kd >> 4) + 4) & ~1;                      // entries (even: 16-byte pairs)
  int2* ttab = (int2*)dyn_lds;
  float* ptab = (float*)(dyn_lds + 2 * ntab);
  // (the first A stage is requested before anything else: it needs no table, and the prologue is two memory round trips
  //  otherwise -- tables, then operands)
  const int nch16_ = g.Kd >> 4;
  const int cbase0 = g.chunks_per_split > 0 ? split * g.chunks_per_split : 0;
  const bool any_chunk = g.chunks_per_split > 0 ? (cbase0 < ((nch16_ + 1) >> 1)) : (nch16_ > 0);
  const unsigned voffA = (unsigned)(lane * 16) + (unsigned)m0 * 64u;
  const unsigned lds_base = (unsigned)(unsigned long)(lds_void)smem;
  if (any_chunk) {
    const unsigned chunk_off = (unsigned)cbase0 * 3u * (unsigned)Mp * 64u;
#pragma unroll
    for (int t = 0; t < (NDMA + WAVES - 1) / WAVES; ++t) {
      const int idx = wave + WAVES * t;
      if (idx < NDMA && SLV_X3_ABL != 2) {
        const int p = idx / MT, rt = idx - p * MT;
        x3_dma16(__builtin_amdgcn_readfirstlane(lds_base + (unsigned)(p * (BM * 64) + rt * 1024)), voffA, rA,
                 __builtin_amdgcn_readfirstlane(chunk_off + (unsigned)(p * Mp + rt * 16) * 64u));
      }
    }
  }
  for (int i = tid; i < ntab; i += NTHR) ttab[i] = g.tab[i];
  if constexpr (PRO == PRO_ACT) {
    for (int i = tid; i < 2 * CbP; i += NTHR) {
      const int which = i >= CbP, c = i - which * CbP;
      ptab[i] = c < g.Cb ? g.pb[which * g.Cb + c] : 0.f;
    }
  }
  __syncthreads();

  // ---- B: this lane's column of each of the wave's NT column tiles
  const int my_tapd = g.tapd[lane];
  unsigned lbase[NT], mlo[NT], mhi[NT];
  size_t obase[NT];                      // where this lane's column of tile j goes in the output (epilogue)
  bool cok[NT];
  const int dP = g.D0 * g.D1 * g.D2;
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const long long n = n0 + (wave * NT + j) * 16 + fi;
    const bool nvalid = n < g.Ntot;
    const unsigned nn = nvalid ? (unsigned)n : 0u;             // (columns < 2^31: the tensors are addressed with 32-bit offsets)
    const unsigned NQ = (unsigned)(g.Q0 * g.Q1 * g.Q2), Q12 = (unsigned)(g.Q1 * g.Q2);
    const unsigned bu = nn / NQ;
    unsigned remu = nn - bu * NQ;
    const unsigned q0u = remu / Q12;
    remu -= q0u * Q12;
    const unsigned q1u = remu / (unsigned)g.Q2;
    const int b = (int)bu, q0 = (int)q0u, q1 = (int)q1u, q2 = (int)(remu - q1u * (unsigned)g.Q2);
    cok[j] = nvalid;
    obase[j] = (size_t)b * g.M * dP + (size_t)(q0 * g.dmul0 + g.dorg0) * (g.D1 * g.D2) +
               (size_t)(q1 * g.dmul1 + g.dorg1) * g.D2 + (size_t)(q2 * g.dmul2 + g.dorg2);
    const int c0 = q0 * g.mul0, c1 = q1 * g.mul1, c2 = q2 * g.mul2;
    lbase[j] = (unsigned)((long long)b * g.sbatch + (long long)c0 * (g.S1 * g.S2) + c1 * g.S2 + c2) + (unsigned)cof * (g.sprod4 >> 2);
    unsigned lo = 0, hi = 0;
    // (no branch on nvalid around this loop: v_readlane reads a lane whether or not it is active, and inside a divergent
    //  region the table load may be sunk into it, leaving the inactive lanes' register unwritten)
    for (int t = 0; t < g.ntaps; ++t) {
      const int d = __builtin_amdgcn_readlane(my_tapd, t);      // (one vector load of the 64-entry table up front: a scalar
                                                                //  load per tap put 9 dependent round trips into the prologue)
      const bool ok = (unsigned)(c0 + (d & 255) - 64) < (unsigned)g.S0 &&
                      (unsigned)(c1 + ((d >> 8) & 255) - 64) < (unsigned)g.S1 &&
                      (unsigned)(c2 + ((d >> 16) & 255) - 64) < (unsigned)g.S2;
      if (t < 32) lo |= (ok ? 1u : 0u) << t;
      else hi |= (ok ? 1u : 0u) << (t - 32);
    }
    mlo[j] = nvalid ? lo : 0u;
    mhi[j] = nvalid ? hi : 0u;
  }
  // ---- chunk range (32-deep chunks; split-K slices in units of them)
  const int nch16 = g.Kd >> 4;
  int nchunks = (nch16 + 1) >> 1, cbase = 0;
  if (g.chunks_per_split > 0) {
    const int c0 = split * g.chunks_per_split;
    int c1 = c0 + g.chunks_per_split;
    if (c1 > nchunks) c1 = nchunks;
    nchunks = c1 > c0 ? c1 - c0 : 0;
    cbase = c0;
  }

  // ---- A: LDS-DMA of this block's rows of the x3 image.  Piece idx = wave + 4 t: plane idx / MT, 16-row tile idx % MT;
  // a lane moves 16 bytes: memory (chunk, plane, row m0 + 16 rt + lane / 4, slot lane % 4) -> the same place of the LDS stage
  auto dma_a = [&](int c) __attribute__((always_inline)) {
    const unsigned chunk_off = (unsigned)(cbase + c) * 3u * (unsigned)Mp * 64u;
    const unsigned dst = lds_base + (unsigned)((c & 1) * A_BYTES);
#pragma unroll
    for (int t = 0; t < (NDMA + WAVES - 1) / WAVES; ++t) {
      const int idx = wave + WAVES * t;
      if (idx < NDMA && SLV_X3_ABL != 2) {
        const int p = idx / MT, rt = idx - p * MT;
        x3_dma16(__builtin_amdgcn_readfirstlane(dst + (unsigned)(p * (BM * 64) + rt * 1024)), voffA, rA,
                 __builtin_amdgcn_readfirstlane(chunk_off + (unsigned)(p * Mp + rt * 16) * 64u));
      }
    }
  };

  // Software pipeline, two chunks deep for B: during iteration c the raw elements of chunk c + 2 are in flight, those of
  // chunk c + 1 are cut into their bf16 pieces BETWEEN the MFMAs of chunk c (a 16x16x32 MFMA occupies the matrix pipe for 16
  // cycles and the wave's issue port for 4: the ~140 VALU instructions of the split ride in those gaps instead of standing
  // between two MFMA phases), and the fragments of chunk c are being multiplied.  Two register sets, indexed by the
  // chunk's parity (the loop is unrolled by two so that every index is a compile-time constant).
  float rb[2][NT][8];                    // raw elements
  int ptoff[2] = {0, 0};                 // PRO_ACT: this lane's 8 channels in the table
  float lim_lo[2][NT], lim_hi[2][NT];    // PRO_ACT: the activation is clamp(x*s + h, lo, hi): (0, inf) with ReLU,
                                         // (-inf, inf) without, (0, 0) where the tap leaves the tensor (zero padding)
  bf16x8 bfr[2][NT][NP];                 // B fragments (NP planes)
  (void)ptoff; (void)lim_lo; (void)lim_hi;

  auto load_chunk = [&](int c, auto par) __attribute__((always_inline)) {
    constexpr int P = decltype(par)::value;
    const int c32 = cbase + c;
    const u32x4 ee = *(const u32x4*)(ttab + 2 * c32);             // one entry per 16-deep group (invalid pad entries behind)
    const int ex = (int)(sc ? ee[2] : ee[0]), ey = (int)(sc ? ee[3] : ee[1]);
    const int tap = ey & 63;
    if constexpr (PRO == PRO_ACT) ptoff[P] = (ey >> 8) + cof;
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const unsigned half = tap < 32 ? mlo[j] : mhi[j];
      const bool ok = (half >> (tap & 31)) & 1u;
      const unsigned voff = ok ? ((lbase[j] + (unsigned)ex) << 2) : OOB;
      if constexpr (PRO == PRO_ACT) {
        lim_hi[P][j] = ok ? __builtin_inff() : 0.f;
        lim_lo[P][j] = (ok && !g.b_relu) ? -__builtin_inff() : 0.f;
      }
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        if (SLV_X3_ABL == 1) rb[P][j][q] = (float)(q + c);
        else rb[P][j][q] = bload_s(rB, voff, (unsigned)q * g.sprod4);
      }
    }
  };

  // raw set P -> fragment set P
  auto split_chunk = [&](auto par) __attribute__((always_inline)) {
    constexpr int P = decltype(par)::value;
    float s8[8], h8[8];
    if constexpr (PRO == PRO_ACT) {
      const f32x4 s0 = *(const f32x4*)(ptab + ptoff[P]), s1 = *(const f32x4*)(ptab + ptoff[P] + 4);
      const f32x4 h0 = *(const f32x4*)(ptab + CbP + ptoff[P]), h1 = *(const f32x4*)(ptab + CbP + ptoff[P] + 4);
#pragma unroll
      for (int q = 0; q < 4; ++q) { s8[q] = s0[q]; s8[4 + q] = s1[q]; h8[q] = h0[q]; h8[4 + q] = h1[q]; }
    }
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      if constexpr (NP == 2) {
        float v8[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          v8[q] = rb[P][j][q];
          if constexpr (PRO == PRO_ACT) v8[q] = __builtin_amdgcn_fmed3f(v8[q] * s8[q] + h8[q], lim_lo[P][j], lim_hi[P][j]);
        }
        unsigned hp[4], mp[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) split2_pair(v8[2 * q], v8[2 * q + 1], hp[q], mp[q]);
        bfr[P][j][0] = __builtin_bit_cast(bf16x8, (u32x4){hp[0], hp[1], hp[2], hp[3]});
        bfr[P][j][1] = __builtin_bit_cast(bf16x8, (u32x4){mp[0], mp[1], mp[2], mp[3]});
      } else {
      unsigned h[8], m[8], l[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        float v = rb[P][j][q];
        if constexpr (PRO == PRO_ACT) v = __builtin_amdgcn_fmed3f(v * s8[q] + h8[q], lim_lo[P][j], lim_hi[P][j]);
        if (SLV_X3_ABL == 3) { h[q] = m[q] = l[q] = __float_as_uint(v); }
        else split3(v, h[q], m[q], l[q]);
      }
      bfr[P][j][0] = __builtin_bit_cast(bf16x8, (u32x4){pack_hi16(h[0], h[1]), pack_hi16(h[2], h[3]), pack_hi16(h[4], h[5]), pack_hi16(h[6], h[7])});
      bfr[P][j][1] = __builtin_bit_cast(bf16x8, (u32x4){pack_hi16(m[0], m[1]), pack_hi16(m[2], m[3]), pack_hi16(m[4], m[5]), pack_hi16(m[6], m[7])});
      bfr[P][j][NP - 1] = __builtin_bit_cast(bf16x8, (u32x4){pack_hi16(l[0], l[1]), pack_hi16(l[2], l[3]), pack_hi16(l[4], l[5]), pack_hi16(l[6], l[7])});
      }
    }
  };

  f32x4 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int mtv = __builtin_amdgcn_readfirstlane((mrem + 15) / 16);
  const int foff = fi * 64 + ((fk ^ cl_swz(fi)) << 4);     // this lane's fragment inside a 16-row tile of a plane

  // One chunk of MFMAs on fragment set P; WITH_SPLIT: the raw set P ^ 1 becomes fragment set P ^ 1 in the same scheduling
  // region.  The A fragments of row-tile step s + 1 are requested BEFORE the MFMAs of step s (left alone, hipcc re-uses two
  // fragment registers and puts a full LDS round trip in front of every MFMA pair); the region's order is laid down with
  // sched_group_barrier: per step 3 G LDS reads, then its 6 G NT MFMAs each followed by two VALU instructions.
  // A step = G row tiles (G = 2 when NT == 1), so that two independent accumulators alternate in the MFMA stream.
  auto compute = [&](int buf, auto par, auto with_split, auto full_tag, int cnext = 0) __attribute__((always_inline)) {
    constexpr int P = decltype(par)::value;
    constexpr bool SPLIT = decltype(with_split)::value;
    constexpr bool FULL = decltype(full_tag)::value;
    constexpr int G = NT == 1 ? 2 : 1;
    constexpr int NS = (MT + G - 1) / G;
    const unsigned char* As = smem + buf * A_BYTES + foff;
    bf16x8 a[2][G][NP];
    __builtin_amdgcn_sched_barrier(0);
#ifdef SLV_X3_PRIO
    __builtin_amdgcn_s_setprio(SLV_X3_PRIO);      // the wave in its MFMA stream wins the SIMD's arbitration: the co-resident wave
#endif                                          // does its loads / split meanwhile instead of both hitting the matrix pipe in phase
#pragma unroll
    for (int gi = 0; gi < G; ++gi)
#pragma unroll
      for (int p = 0; p < NP; ++p) a[0][gi][p] = *(const bf16x8*)(As + (gi < MT ? gi : 0) * 1024 + p * BM * 64);
    if constexpr (SPLIT && FULL && FUSE) split_chunk(std::integral_constant<int, P ^ 1>{});
    (void)cnext;
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      if (s + 1 < NS) {
#pragma unroll
        for (int gi = 0; gi < G; ++gi) {
          const int i = (s + 1) * G + gi < MT ? (s + 1) * G + gi : MT - 1;
#pragma unroll
          for (int p = 0; p < NP; ++p) a[(s + 1) & 1][gi][p] = *(const bf16x8*)(As + i * 1024 + p * BM * 64);
        }
      }
      if constexpr (!FULL || !FUSE) __builtin_amdgcn_sched_barrier(0);
      // products (plane of A, plane of B), small terms first; the G x NT accumulators of the step alternate
      constexpr int PA[6] = {NP == 3 ? 2 : 1, 0, NP == 3 ? 1 : 0, 1, 0, 0}, PB[6] = {0, NP == 3 ? 2 : 1, NP == 3 ? 1 : 0, 0, 1, 0};
#pragma unroll
      for (int t = 0; t < NPROD; ++t)
#pragma unroll
        for (int gi = 0; gi < G; ++gi) {
          const int i = s * G + gi;
          if (i < MT && (FULL || i < mtv)) {
#pragma unroll
            for (int j = 0; j < NT; ++j) {
              if (SLV_X3_ABL == 4) { if (t == 0) acc[i][j][0] += (float)a[s & 1][gi][0][0] + (float)a[s & 1][gi][1][0] + (float)a[s & 1][gi][NP - 1][0] + (float)bfr[P][j][0][0] + (float)bfr[P][j][1][0] + (float)bfr[P][j][NP - 1][0]; }
              else acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[s & 1][gi][PA[t]], bfr[P][j][PB[t]], acc[i][j], 0, 0, 0);
            }
          }
        }
      if constexpr (!FULL || !FUSE) __builtin_amdgcn_sched_barrier(0);
    }
    if constexpr (FULL && FUSE) {
      // the order of the region: [reads of step 0 (+ the table reads of the split)] then per step [reads of step s + 1]
      // [its MFMAs, two VALU behind each]
      __builtin_amdgcn_sched_group_barrier(0x100, NP * G + (SPLIT && PRO == PRO_ACT ? 4 : 0), 0);
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        if (s + 1 < NS) __builtin_amdgcn_sched_group_barrier(0x100, NP * G, 0);
#pragma unroll
        for (int t = 0; t < NPROD * G * NT; ++t) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          if (SPLIT) __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
#ifdef SLV_X3_PRIO
      __builtin_amdgcn_s_setprio(0);
#endif
    } else {
#ifdef SLV_X3_PRIO
      __builtin_amdgcn_s_setprio(0);
#endif
      if constexpr (SPLIT) split_chunk(std::integral_constant<int, P ^ 1>{});
    }
  };

  auto main_loop = [&](auto full_tag) __attribute__((always_inline)) {
    constexpr std::integral_constant<int, 0> P0{};
    constexpr std::integral_constant<int, 1> P1{};
    X3_T(1);
    load_chunk(0, P0);                                    // (stage 0 was requested at the top of the kernel)
    if (FUSE && nchunks > 1) load_chunk(1, P1);
    asm volatile("s_waitcnt vmcnt(0)");                   // this wave's DMA pieces (the compiler does not count them)
    __syncthreads();                 // ... everybody's
    split_chunk(P0);
    X3_T(2);
    if (PP && wave >= 4) x3_pp_barrier();                 // the second wave group runs one slot behind the first from here on
    // iteration c: stage / fragment set c & 1 is multiplied, raw set (c + 1) & 1 is split, raw set c & 1 is re-loaded (chunk c + 2)
    // (!FUSE: raw set (c + 1) & 1 is loaded at the top of the iteration and split at its end)
    auto iter = [&](int c, auto par) __attribute__((always_inline)) {
      constexpr int P = decltype(par)::value;
#ifdef SLV_X3_TRACE
      const int tb = 8 + 8 * (c < 17 ? c : 17);
#endif
      X3_T(tb + 0);
      dma_a(c + 1);                  // into the stage every wave finished reading before the previous barrier
      X3_T(tb + 1);
      if constexpr (FUSE) {
        if (c + 2 < nchunks) load_chunk(c + 2, par);       // (placing these loads between the MFMAs as well: measured, no gain)
      } else {
        load_chunk(c + 1, std::integral_constant<int, P ^ 1>{});
      }
      X3_T(tb + 2);
      if constexpr (PP) {
        __builtin_amdgcn_s_setprio(1);
        compute(P, par, std::false_type{}, full_tag);      // M slot: the MFMAs of chunk c, nothing else
        __builtin_amdgcn_s_setprio(0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's DMA pieces and raw elements of chunk c + 1
        x3_pp_barrier();
        split_chunk(std::integral_constant<int, P ^ 1>{}); // S slot
        x3_pp_barrier();
      } else {
        compute(P, par, std::true_type{}, full_tag, c + 2);  // (!FUSE: the split of raw set P ^ 1 follows the MFMAs inside)
        X3_T(tb + 3);
        asm volatile("s_waitcnt vmcnt(0)");                 // this wave's DMA pieces of chunk c + 1 (the barrier's fence pins the order)
        __syncthreads();               // stage (c + 1) & 1 has landed for everybody and stage c & 1 is free
      }
      X3_T(tb + 5);
    };
    int c = 0;
    for (; c + 2 < nchunks; c += 2) {
      iter(c, P0);
      iter(c + 1, P1);
    }
    if (c + 1 < nchunks) {           // an even number of chunks: one more iteration, then the last chunk (odd parity)
      iter(c, P0);
      compute(1, P1, std::false_type{}, full_tag);
    } else {
      compute(0, P0, std::false_type{}, full_tag);
    }
    if (PP && wave < 4) x3_pp_barrier();                  // (the barrier the second group was given in front of the loop)
    __syncthreads();
    X3_T(3);
  };
  if (nchunks > 0) {
    if (mtv >= MT) main_loop(std::true_type{});
    else main_loop(std::false_type{});
  } else {
    __syncthreads();
  }

  // ---------------------------------------------------------------- epilogue (as igemm.hpp MODE_CONV, 16 x 16 tiles)
  // accumulator layout: row = i*16 + fk*4 + r, col = (wave*NT + j)*16 + fi
  float* Cp = g.C + (size_t)split * (size_t)g.split_stride;
  float* red = (float*)smem;            // [2][WAVES][BM]   (the main loop ended with a barrier)
  if constexpr (EPI == EPI_PLAIN) {
    auto store_all = [&](auto has_e) __attribute__((always_inline)) {
      constexpr bool HAS_E = decltype(has_e)::value;
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        if (i < mtv) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int m = i * 16 + fk * 4 + r;
            if (m < mrem) {
#pragma unroll
              for (int j = 0; j < NT; ++j) {
                if (cok[j]) {
                  const size_t ad = obase[j] + (size_t)(m0 + m) * dP;
                  float v = acc[i][j][r];
                  if constexpr (HAS_E) v += g.E[ad];
                  Cp[ad] = v;
                }
              }
            }
          }
        }
      }
    };
    if (g.E) store_all(std::true_type{});      // (one wave-uniform branch, not one per element)
    else store_all(std::false_type{});
  } else if constexpr (EPI == EPI_EVAL) {
    // y = relu?(acc + bias[row] (+ E)): conv + folded BatchNorm (+ residual) + ReLU of a torchvision block (model.py:95,114)
    auto store_all = [&](auto has_e) __attribute__((always_inline)) {
      constexpr bool HAS_E = decltype(has_e)::value;
      const float lo = g.epi_relu ? 0.f : -__builtin_inff();
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        if (i < mtv) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int m = i * 16 + fk * 4 + r;
            if (m < mrem) {
              const float bv = g.bias ? g.bias[m0 + m] : 0.f;
#pragma unroll
              for (int j = 0; j < NT; ++j) {
                if (cok[j]) {
                  const size_t ad = obase[j] + (size_t)(m0 + m) * dP;
                  float v = acc[i][j][r] + bv;
                  if constexpr (HAS_E) v += g.E[ad];
                  Cp[ad] = fmaxf(v, lo);
                }
              }
            }
          }
        }
      }
    };
    if (g.E) store_all(std::true_type{});
    else store_all(std::false_type{});
  } else {
    float* rpar = red + 2 * WAVES * BM;  // [4][BM] s, h, mean, invstd of this block's rows
    for (int m = tid; m < BM; m += NTHR) {
      const int mm = (m0 + m < g.M) ? m0 + m : g.M - 1;
      rpar[m] = g.rss[mm];
      rpar[BM + m] = g.rss[g.M + mm];
      rpar[2 * BM + m] = g.rmi[mm];
      rpar[3 * BM + m] = g.rmi[g.M + mm];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      if (i < mtv) {
        float xv[4][NT], ev[4][NT];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int m = i * 16 + fk * 4 + r;
#pragma unroll
          for (int j = 0; j < NT; ++j) {
            const bool ok = (m < mrem) && cok[j];
            const size_t ad = obase[j] + (size_t)(m0 + m) * dP;
            xv[r][j] = ok ? g.R[ad] : 0.f;
            ev[r][j] = (g.E && ok) ? g.E[ad] : 0.f;
          }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int m = i * 16 + fk * 4 + r;
          const bool mok = m < mrem;
          const float ps = rpar[m], ph = rpar[BM + m], pm = rpar[2 * BM + m], pi = rpar[3 * BM + m];
          float s0 = 0.f, s1 = 0.f;
#pragma unroll
          for (int j = 0; j < NT; ++j) {
            if (mok && cok[j]) {
              const size_t ad = obase[j] + (size_t)(m0 + m) * dP;
              const float v = acc[i][j][r] + ev[r][j];
              Cp[ad] = v;
              const float gm = (xv[r][j] * ps + ph > 0.f) ? v : 0.f;
              s0 += gm;
              s1 += gm * ((xv[r][j] - pm) * pi);
            }
          }
          s0 = row16_sum(s0);      // the 16 lanes of a k group are a DPP row (ds_bpermute shuffles cost an LDS round trip each:
          s1 = row16_sum(s1);      // 288 of them were 10 % of a workgroup's life)
          if (fi == 0) {
            red[wave * BM + m] = s0;
            red[(WAVES + wave) * BM + m] = s1;
          }
        }
      }
    }
    __syncthreads();
    for (int m = tid; m < BM; m += NTHR) {
      if (m < mrem) {
        float a = 0.f, b = 0.f;
#pragma unroll
        for (int w = 0; w < WAVES; ++w) { a += red[w * BM + m]; b += red[(WAVES + w) * BM + m]; }
        float* o = g.rpart + ((size_t)(m0 + m) * g.rslots + g.rslot0 + nblk) * 2;
        o[0] = a;
        o[1] = b;
      }
    }
  }
  if (g.stat_sum) {
    if constexpr (EPI != EPI_PLAIN) __syncthreads();
#pragma unroll
    for (int i = 0; i < MT; ++i) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float s = 0.f, q = 0.f;
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          const float v = cok[j] ? acc[i][j][r] : 0.f;
          s += v;
          q += v * v;
        }
        s = row16_sum(s);
        q = row16_sum(q);
        if (fi == 0) {
          const int m = i * 16 + fk * 4 + r;
          red[wave * BM + m] = s;
          red[(WAVES + wave) * BM + m] = q;
        }
      }
    }
    __syncthreads();
    for (int m = tid; m < BM; m += NTHR) {
      if (m < mrem) {
        float s = 0.f, q = 0.f;
#pragma unroll
        for (int w = 0; w < WAVES; ++w) { s += red[w * BM + m]; q += red[(WAVES + w) * BM + m]; }
        g.stat_sum[(size_t)(m0 + m) * g.nblkN + nblk] = s;
        g.stat_sq[(size_t)(m0 + m) * g.nblkN + nblk] = q;
      }
    }
  }
  X3_T(4);
}

// NT_BLK: block columns / 64 (the launch configuration's nt): 1, 2 -> 4 waves of NT_BLK column tiles; 4 -> 8 waves of 2
template <int MT, int NT_BLK, int SUBSET>
inline void launch_igemm3(IgemmArgs a, int splits, hipStream_t st) {
  constexpr int WAVES = NT_BLK >= 4 ? 8 : 4, NT = NT_BLK >= 4 ? NT_BLK / 2 : NT_BLK;
  dim3 grid(a.nblkM * a.nblkN * splits, 1, 1);
  if (a.chunks_per_split > 0) {   // the caller counted 16-deep chunks
    const int ch32 = ((a.Kd >> 4) + 1) >> 1;
    a.chunks_per_split = (ch32 + splits - 1) / splits;
  }
  const bool act = a.b_pro == PRO_ACT;
  // workgroups per CU the register budget is cut for: the LDS (two A stages) admits 2 at MT = 9, 3 at MT = 8, 6 at MT = 4
  // MT = 9 (two A stages = 55 KB: two workgroups per CU at most): fused two-deep pipeline; MT <= 8: the lean pipeline, three
  // workgroups per CU (the fused one spills under that register cap)
  // (-DSLV_X3_PP=1: the 8-wave tiles on the lean pipeline in its ping-pong form, one workgroup per CU = two waves per SIMD.
  //  Measured (tools/x3_pp_ab.py, profiles/r05_notes.md): SLOWER than the free-running 4-wave forms on every layer -- 159 vs 192
  //  TFLOP/s on the layer-1 spatial forward: with the slots fenced by barriers nothing may fill the matrix pipe while the wave
  //  that owns the M slot waits for its next A fragments, one LDS round trip per row-tile step.  Off.)
#ifndef SLV_X3_PP
#define SLV_X3_PP 0
#endif
  constexpr bool FUSE_ = (WAVES == 8 && !SLV_X3_PP) || (WAVES != 8 && MT >= 9);
  constexpr int OCC_ = WAVES == 8 ? (SLV_X3_PP ? 2 : 1) : (FUSE_ ? 2 : 3);
  const size_t dyn = (size_t)((((a.Kd >> 4) + 4) & ~1) * 8) + (act ? (size_t)2 * ((a.Cb + 15) / 16 * 16) * sizeof(float) : 0);
#define SLV_K3(PRO_, EPI_) \
  hipLaunchKernelGGL((igemm3_kernel<MT, NT, PRO_, EPI_, WAVES, OCC_, FUSE_>), grid, dim3(64 * WAVES), dyn, st, a)
  if constexpr (SUBSET != SUB_FWD) {
    if (a.R) { SLV_K3(PRO_NONE, EPI_BNR); return; }
  }
  if constexpr (SUBSET != SUB_DGRAD) {
    if (act) { SLV_K3(PRO_ACT, EPI_PLAIN); return; }
  }
  SLV_K3(PRO_NONE, EPI_PLAIN);
#undef SLV_K3
}

// eval-mode forward (EPI_EVAL, no prologue, unsplit K), NP pieces per operand
template <int MT, int NT_BLK, int NP>
inline void launch_igemm3_eval(IgemmArgs a, hipStream_t st) {
  constexpr int WAVES = NT_BLK >= 4 ? 8 : 4, NT = NT_BLK >= 4 ? NT_BLK / 2 : NT_BLK;
  dim3 grid(a.nblkM * a.nblkN, 1, 1);
  a.chunks_per_split = 0;
  constexpr bool FUSE_ = WAVES == 8 || MT >= 9;
  constexpr int OCC_ = WAVES == 8 ? 1 : ((FUSE_ || MT * NT >= 16) ? 2 : 3);      // (the 128 x 128 lean tile spills under the 3-per-CU cap)
  const size_t dyn = (size_t)((((a.Kd >> 4) + 4) & ~1) * 8);
  hipLaunchKernelGGL((igemm3_kernel<MT, NT, PRO_NONE, EPI_EVAL, WAVES, OCC_, FUSE_, NP>), grid, dim3(64 * WAVES), dyn, st, a);
}

}  // namespace slv

namespace slv {

// ---------------------------------------------------------------------------------------------------------------------
// Weight gradient with split operands:  dW[co][(ci,tap)] = sum_p dXout[co][p] * act(X)[ci][p*s + tap - pad]
// (igemm.hpp MODE_WGRAD with 16-byte loads of BOTH operands, its VA + VB form: stride-1 "same" spatial taps or 1 x 1
// spatial extent, To*Ho*Wo % 4 == 0).  The contraction index is the position, contiguous in memory for both operands:
// a thread loads quads of 4 consecutive positions (a K chunk = 32 positions = 8 quads), splits them into the three bf16
// planes and writes 8-byte pieces of the [row][32 k] LDS rows.  Deterministic split-K over positions as before.
// VB: how the gathered operand is read -- 2: quads of 4 consecutive positions = 4 consecutive input elements of one row
// (stride-1 "same" spatial taps, Wo % 4 == 0; the quad at the image border is loaded aligned and shifted), 1: the same
// without spatial taps (temporal / pointwise convs, Ho*Wo % 4 == 0), 0: element by element (strided spatial convs, 7 x 7 and
// 14 x 14 maps: any geometry).  VA: the gradient operand by 16-byte loads (To*Ho*Wo % 4 == 0) or element by element.
template <int MT, int NT, int PRO, int VB, bool VA, int OCC>
__global__ __launch_bounds__(256, OCC) void igemm3_wgrad_kernel(const IgemmArgs g) {
  constexpr int BM = MT * 16, BN = NT * 64;
  constexpr int NPA = (BM + 31) / 32;
  constexpr int A_BYTES = 3 * BM * 64;
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * A_BYTES];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int mblk, nblk, split;
  {
    const int nb = gridDim.x, id = blockIdx.x;
    const int q = nb >> 3, r = nb & 7, xcd = id & 7, loc = id >> 3;
    const int nid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
    const int per = g.nblkM * g.nblkN;
    split = nid / per;
    const int rem_ = nid - split * per;
    mblk = rem_ % g.nblkM;
    nblk = rem_ / g.nblkM;
  }
  const int m0 = mblk * BM;
  const long long n0 = (long long)nblk * BN;
  const int mrem = g.M - m0;
  const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc((void*)g.A, 0, (int)g.A_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rB = __builtin_amdgcn_make_buffer_rsrc((void*)g.B, 0, (int)g.B_bytes, 0x00020000);

  const int aq = tid & 7, ar = tid >> 3;            // A loader: position quad aq of the chunk, row ar + 32 i
  const int fi = lane & 15, fk = lane >> 4;         // fragment lane: column fi of a tile, positions 8 fk .. 8 fk + 7
  const int HWi = g.Hi * g.Wi, THWi = g.Ti * HWi;
  const int HoWo = g.Ho * g.Wo, Pout = g.To * HoWo;
  int nchunks;
  unsigned pbase;      // first position of the slice
  {
    const long long c0 = (long long)split * g.chunks_per_split;
    const long long call = (g.Ptot + 31) / 32;
    long long c1 = c0 + g.chunks_per_split;
    if (c1 > call) c1 = call;
    nchunks = (int)(c1 > c0 ? c1 - c0 : 0);
    pbase = (unsigned)(c0 * 32);
  }
  // B (the gathered activations): a wave's columns are its own -- every lane loads the elements of ITS fragments (column fi
  // of each of the NT column tiles, two quads of consecutive positions) and keeps them in registers: no LDS, no barrier
  int vb_off[NT], vb_dt[NT], vb_dh[NT], vb_dw[NT];
  float vb_s[NT], vb_h[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const long long n = n0 + (wave * NT + j) * 16 + fi;
    const int2 e = g.tab[n < g.Ntot ? n : g.Ntot];          // entry Ntot is an invalid pad entry (tap 63)
    vb_off[j] = e.x;
    const bool pad_entry = (e.y & 63) == 63;
    const int d = pad_entry ? 0 : g.tapd[e.y & 63];
    vb_dt[j] = pad_entry ? (1 << 20) : ((d & 255) - 64);
    vb_dh[j] = ((d >> 8) & 255) - 64;
    vb_dw[j] = ((d >> 16) & 255) - 64;
    const int ch = pad_entry ? 0 : (e.y >> 8);
    vb_s[j] = (PRO == PRO_ACT) ? g.pb[ch] : 1.f;
    vb_h[j] = (PRO == PRO_ACT) ? g.pb[g.Cin + ch] : 0.f;
  }
  const float lo_act = g.b_relu ? 0.f : -__builtin_inff();

  f32x4 ra4[NPA], rb4[NT][2];
  unsigned okB = 0, edgeLo = 0, edgeHi = 0;     // bit j * 2 + quad

  auto load_chunk = [&](int c) __attribute__((always_inline)) {
    if constexpr (VA) {
      const unsigned pq = pbase + (unsigned)c * 32u + 4u * (unsigned)aq;
      const bool qok = (long long)pq < g.Ptot;
      const unsigned pqq = qok ? pq : 0u;
      const unsigned bq = fdiv(pqq, g.dPout);
      const unsigned remq = pqq - bq * (unsigned)Pout;
      const unsigned abase = (bq * (unsigned)g.Cout + (unsigned)(m0 + ar)) * (unsigned)Pout + remq;
#pragma unroll
      for (int i = 0; i < NPA; ++i) {
        const unsigned off = (qok && ar + 32 * i < BM) ? ((abase + (unsigned)(32 * i) * (unsigned)Pout) << 2) : OOB;
        ra4[i] = bload4(rA, off);
      }
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const unsigned pe = pbase + (unsigned)c * 32u + 4u * (unsigned)aq + (unsigned)e;
        const bool pok = (long long)pe < g.Ptot;
        const unsigned pp = pok ? pe : 0u;
        const unsigned bq = fdiv(pp, g.dPout);
        const unsigned abase = (bq * (unsigned)g.Cout + (unsigned)(m0 + ar)) * (unsigned)Pout + (pp - bq * (unsigned)Pout);
#pragma unroll
        for (int i = 0; i < NPA; ++i)
          ra4[i][e] = bload(rA, (pok && ar + 32 * i < BM) ? ((abase + (unsigned)(32 * i) * (unsigned)Pout) << 2) : OOB);
      }
    }
    okB = edgeLo = edgeHi = 0;
    if constexpr (VB == 0) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const unsigned pe = pbase + (unsigned)c * 32u + 8u * (unsigned)fk + (unsigned)e;
        const bool pok = (long long)pe < g.Ptot;
        const unsigned pp = pok ? pe : 0u;
        const unsigned bq = fdiv(pp, g.dPout);
        const unsigned rem = pp - bq * (unsigned)Pout;
        const unsigned to = fdiv(rem, g.dHoWo);
        const unsigned r2 = rem - to * (unsigned)HoWo;
        const unsigned ho = fdiv(r2, g.dWo);
        const unsigned wo = r2 - ho * (unsigned)g.Wo;
        const int ti = (int)to * g.st, hi_ = (int)ho * g.sh, wi = (int)wo * g.sw;      // (the table's deltas include "- pad")
        const unsigned xb = bq * (unsigned)(g.Cin * THWi) + (unsigned)(ti * HWi + hi_ * g.Wi + wi);
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          const bool ok = pok && (unsigned)(ti + vb_dt[j]) < (unsigned)g.Ti && (unsigned)(hi_ + vb_dh[j]) < (unsigned)g.Hi &&
                          (unsigned)(wi + vb_dw[j]) < (unsigned)g.Wi;
          rb4[j][e >> 2][e & 3] = bload(rB, ok ? ((xb + (unsigned)vb_off[j]) << 2) : OOB);
          okB |= (ok ? 1u : 0u) << (8 * j + e);
        }
      }
      return;
    }
#pragma unroll
    for (int qd = 0; qd < 2; ++qd) {
      const unsigned pq = pbase + (unsigned)c * 32u + 8u * (unsigned)fk + 4u * (unsigned)qd;
      const bool qok = (long long)pq < g.Ptot;
      const unsigned pqq = qok ? pq : 0u;
      const unsigned bq = fdiv(pqq, g.dPout);
      const unsigned remq = pqq - bq * (unsigned)Pout;
      const unsigned toq = fdiv(remq, g.dHoWo);
      const unsigned r2q = remq - toq * (unsigned)HoWo;
      unsigned hoq = 0, woq = 4;
      if constexpr (VB == 2) {
        hoq = fdiv(r2q, g.dWo);
        woq = r2q - hoq * (unsigned)g.Wo;
      }
      const unsigned xbq = bq * (unsigned)(g.Cin * THWi) + toq * (unsigned)(g.st * HWi) + r2q;   // Hi = Ho, Wi = Wo
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const bool ok = qok && (unsigned)((int)toq * g.st + vb_dt[j]) < (unsigned)g.Ti &&
                        (VB == 1 || (unsigned)((int)hoq + vb_dh[j]) < (unsigned)g.Hi);
        const bool lo = VB == 2 && vb_dw[j] < 0 && woq == 0u;
        const bool hi = VB == 2 && vb_dw[j] > 0 && woq + 4u == (unsigned)g.Wo;
        const unsigned el = xbq + (unsigned)vb_off[j] + (lo ? 1u : 0u) - (hi ? 1u : 0u);
        rb4[j][qd] = bload4(rB, ok ? (el << 2) : OOB);
        okB |= (ok ? 1u : 0u) << (2 * j + qd);
        edgeLo |= (lo ? 1u : 0u) << (2 * j + qd);
        edgeHi |= (hi ? 1u : 0u) << (2 * j + qd);
      }
    }
  };

  auto put = [&](unsigned char* dst, int plane_bytes, f32x4 v) __attribute__((always_inline)) {
    unsigned h[4], m[4], l[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) split3(v[j], h[j], m[j], l[j]);
    *(u32x2*)(dst) = (u32x2){pack_hi16(h[0], h[1]), pack_hi16(h[2], h[3])};
    *(u32x2*)(dst + plane_bytes) = (u32x2){pack_hi16(m[0], m[1]), pack_hi16(m[2], m[3])};
    *(u32x2*)(dst + 2 * plane_bytes) = (u32x2){pack_hi16(l[0], l[1]), pack_hi16(l[2], l[3])};
  };
  auto store_a = [&](int buf) __attribute__((always_inline)) {
    unsigned char* As = smem + buf * A_BYTES;
#pragma unroll
    for (int i = 0; i < NPA; ++i) {
      const int row = ar + 32 * i;
      if (row < BM) put(As + row * 64 + (((aq >> 1) ^ cl_swz(row & 15)) << 4) + (aq & 1) * 8, BM * 64, ra4[i]);
    }
  };
  // FUSE: the next chunk's split runs between this chunk's MFMAs and needs a second fragment set (NT <= 2; the 192-column
  // tiles have no registers for it: their split stands between two chunks)
  constexpr bool FUSE = NT <= 2;
  bf16x8 bfr[FUSE ? 2 : 1][NT][3];   // the B fragments of the chunk being multiplied (set = its parity) and of the next one
  auto split_b = [&](auto par) __attribute__((always_inline)) {
    constexpr int P = FUSE ? decltype(par)::value : 0;
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      unsigned w[3][4];
#pragma unroll
      for (int qd = 0; qd < 2; ++qd) {
        const int bit = 2 * j + qd;
        const bool lo = (edgeLo >> bit) & 1u, hi = (edgeHi >> bit) & 1u;
        f32x4 q = rb4[j][qd];
        if constexpr (PRO == PRO_ACT) {   // activation BEFORE the border shift: padding is zero after BN + ReLU
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const bool ok = VB == 0 ? ((okB >> (8 * j + 4 * qd + e)) & 1u) : ((okB >> bit) & 1u);
            const float l_ = ok ? lo_act : 0.f, h_ = ok ? __builtin_inff() : 0.f;
            q[e] = __builtin_amdgcn_fmed3f(q[e] * vb_s[j] + vb_h[j], l_, h_);
          }
        }
        const f32x4 sh = lo ? (f32x4){0.f, q[0], q[1], q[2]} : (hi ? (f32x4){q[1], q[2], q[3], 0.f} : q);
        unsigned h[4], m[4], l[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) split3(sh[e], h[e], m[e], l[e]);
        w[0][2 * qd] = pack_hi16(h[0], h[1]); w[0][2 * qd + 1] = pack_hi16(h[2], h[3]);
        w[1][2 * qd] = pack_hi16(m[0], m[1]); w[1][2 * qd + 1] = pack_hi16(m[2], m[3]);
        w[2][2 * qd] = pack_hi16(l[0], l[1]); w[2][2 * qd + 1] = pack_hi16(l[2], l[3]);
      }
#pragma unroll
      for (int p = 0; p < 3; ++p) bfr[P][j][p] = __builtin_bit_cast(bf16x8, (u32x4){w[p][0], w[p][1], w[p][2], w[p][3]});
    }
  };

  f32x4 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const int mtv = __builtin_amdgcn_readfirstlane((mrem + 15) / 16);
  const int foff = fi * 64 + ((fk ^ cl_swz(fi)) << 4);

  // One chunk of MFMAs on stage / fragment set P; WITH_NEXT: the raw elements of the next chunk (requested at the top of the
  // iteration) are cut into their pieces in the SAME scheduling region -- the gradient quads into LDS stage P ^ 1, the
  // gathered quads into fragment set P ^ 1 -- between the MFMAs of the later row-tile steps (the loads have landed by then):
  // ~350 VALU instructions and 15 LDS writes per chunk ride in the MFMAs' issue gaps instead of standing between two chunks.
  auto compute = [&](auto par, auto with_next, auto full_tag) __attribute__((always_inline)) {
    constexpr int P = decltype(par)::value, PF = FUSE ? P : 0;
    constexpr bool NEXT = decltype(with_next)::value;
    constexpr bool FULL = decltype(full_tag)::value;
    constexpr int G = NT == 1 ? 2 : 1;
    constexpr int NS = (MT + G - 1) / G;
    constexpr int S0 = NS >= 6 ? 2 : 1;                     // first step that carries split work
    const unsigned char* As = smem + P * A_BYTES + foff;
    bf16x8 a[2][G][3];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int gi = 0; gi < G; ++gi)
#pragma unroll
      for (int p = 0; p < 3; ++p) a[0][gi][p] = *(const bf16x8*)(As + (gi < MT ? gi : 0) * 1024 + p * BM * 64);
    if constexpr (NEXT && FULL && FUSE) {
      store_a(P ^ 1);
      split_b(std::integral_constant<int, P ^ 1>{});
    }
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      if (s + 1 < NS) {
#pragma unroll
        for (int gi = 0; gi < G; ++gi) {
          const int i = (s + 1) * G + gi < MT ? (s + 1) * G + gi : MT - 1;
#pragma unroll
          for (int p = 0; p < 3; ++p) a[(s + 1) & 1][gi][p] = *(const bf16x8*)(As + i * 1024 + p * BM * 64);
        }
      }
      if constexpr (!FULL || !FUSE) __builtin_amdgcn_sched_barrier(0);
      constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
      for (int t = 0; t < 6; ++t)
#pragma unroll
        for (int gi = 0; gi < G; ++gi) {
          const int i = s * G + gi;
          if (i < MT && (FULL || i < mtv)) {
#pragma unroll
            for (int j = 0; j < NT; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[s & 1][gi][PA[t]], bfr[PF][j][PB[t]], acc[i][j], 0, 0, 0);
          }
        }
      if constexpr (!FULL || !FUSE) __builtin_amdgcn_sched_barrier(0);
    }
    if constexpr (FULL && FUSE) {
      __builtin_amdgcn_sched_group_barrier(0x100, 3 * G, 0);
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        if (s + 1 < NS) __builtin_amdgcn_sched_group_barrier(0x100, 3 * G, 0);
#pragma unroll
        for (int t = 0; t < 6 * G * NT; ++t) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          if (NEXT && s >= S0) {
            __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
            if ((t & 3) == 3) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
          }
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    } else if constexpr (NEXT) {
      store_a(P ^ 1);
      split_b(std::integral_constant<int, P ^ 1>{});
    }
  };
  auto main_loop = [&](auto full_tag) __attribute__((always_inline)) {
    constexpr std::integral_constant<int, 0> P0{};
    constexpr std::integral_constant<int, 1> P1{};
    load_chunk(0);
    store_a(0);
    split_b(P0);
    __syncthreads();
    // iteration c: stage / fragment set c & 1 is multiplied; the raw elements of chunk c + 1 are loaded at its top and cut
    // into stage / set (c + 1) & 1 inside it (every wave finished reading that stage before the previous barrier)
    auto iter = [&](int c, auto par) __attribute__((always_inline)) {
      load_chunk(c + 1);
      compute(par, std::true_type{}, full_tag);
      __syncthreads();
    };
    int c = 0;
    for (; c + 2 < nchunks; c += 2) {
      iter(c, P0);
      iter(c + 1, P1);
    }
    if (c + 1 < nchunks) {
      iter(c, P0);
      compute(P1, std::false_type{}, full_tag);
    } else {
      compute(P0, std::false_type{}, full_tag);
    }
  };
  if (nchunks > 0) {
    if (mtv >= MT) main_loop(std::true_type{});
    else main_loop(std::false_type{});
  }
  float* Cout_ = g.C + (size_t)split * g.M * g.ldc;
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    if (i < mtv) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = i * 16 + fk * 4 + r;
        if (m < mrem) {
#pragma unroll
          for (int j = 0; j < NT; ++j) {
            const long long n = n0 + (wave * NT + j) * 16 + fi;
            if (n < g.Ntot) Cout_[(size_t)(m0 + m) * g.ldc + n] = acc[i][j][r];
          }
        }
      }
    }
  }
}

template <int MT, int NT>
inline void launch_igemm3_wgrad(IgemmArgs a, int splits, bool vec_a, hipStream_t st) {
  dim3 grid(a.nblkM * a.nblkN * splits, 1, 1);
  const long long ch32 = (a.Ptot + 31) / 32;
  a.chunks_per_split = (int)((ch32 + splits - 1) / splits);
  const bool act = a.b_pro == PRO_ACT;
#ifndef SLV_X3_WOCC
#define SLV_X3_WOCC 2
#endif
#define SLV_W3(PRO_, VB_, VA_) \
  hipLaunchKernelGGL((igemm3_wgrad_kernel<MT, NT, PRO_, VB_, VA_, SLV_X3_WOCC>), grid, dim3(256), 0, st, a)
  const int vb = vec_a ? a.vec_b : 0;        // (the quad forms need the 16-byte gradient loads as well)
  if (vb == 1) { if (act) SLV_W3(PRO_ACT, 1, true); else SLV_W3(PRO_NONE, 1, true); }
  else if (vb == 2) { if (act) SLV_W3(PRO_ACT, 2, true); else SLV_W3(PRO_NONE, 2, true); }
  else if (vec_a) { if (act) SLV_W3(PRO_ACT, 0, true); else SLV_W3(PRO_NONE, 0, true); }
  else { if (act) SLV_W3(PRO_ACT, 0, false); else SLV_W3(PRO_NONE, 0, false); }
#undef SLV_W3
}

}  // namespace slv
