// 16-bit MFMA path: the stride-1 (1,3,3) "same" convolutions -- two thirds of the video trunk's FLOPs, forward AND
// backward data (a stride-1 backward-data conv is the same conv on flipped taps) -- with the input PATCH resident in LDS.
//
// csrc/conv_cl16.hip fetches every activation row once per tap (9x) and the whole weight slab once per 128 positions: the
// kernel is bound by its L2 -> LDS traffic (17.4 KB per K-step and block), not by MFMA or HBM.  Here a block of 8 waves
// owns 256 CONSECUTIVE positions (flat index over n,t,h,w) and, per group of 64 (or 32) input channels, loads the flat
// run of rows [p0 - W - 1, p0 + 256 + W + 1) ONCE into LDS: every tap (dh, dw) is then the same LDS image read at a row
// offset dh*W + dw, with the lanes whose tap falls outside the image (w + dw, h + dh out of range -- the flat run wraps
// into the neighbouring image row / frame there) zeroed in registers.  Only the weights stream per tap.  Per 256
// positions and 64 channels: 47 KB of activations + 9 x 18 KB of weights against 2 x 9 x 2 x 17.4 KB = 626 KB before;
// the train-mode prologue (BatchNorm + ReLU on load) runs once per element instead of once per tap.
//
// Pipeline: the patch of channel group g+1 is fetched into registers at the start of group g and written to LDS at its
// end (9 stages later); the weights of the next (tap, group) stage are fetched one stage ahead
// (register-staged double buffer, one barrier per stage = per 2 K-steps of 32 channels).  One patch buffer: the next
// group's rows are written behind an extra barrier after the group's last tap.  Blocks of SLV_S3_NW waves (32 positions each).
// Epilogue as in conv_cl16.hip (transposed tile through LDS, 16-byte stores, optional statistics / affine / residual).
#include "cl16.hpp"
#include "../../include/selavi_hip.h"

namespace slv {

#ifndef SLV_S3_NW
#define SLV_S3_NW 4
#endif
constexpr int S3_NW = SLV_S3_NW, S3_BN = 32 * S3_NW, S3_THREADS = 64 * S3_NW;
constexpr int S3_PIT = (S3_BN + 2 * 56 + 2) * 8 / S3_THREADS + 1;      // S3_PIT * threads >= patch rows * 8 (W <= 56)
#ifndef SLV_S3_ABL
#define SLV_S3_ABL 0     // timing ablations (tools/s3_ablate.sh): 1 no MFMA, 2 no fragment reads, 3 no weight loads,
#endif                   // 4 no output stores, 5 no barriers in the K loop, 6 no patch loads, 7 no K loop at all, 8 the K loop
                         // twice -- results are wrong for != 0.  Measured on the layer-1 forward (0.284 ms): 7 -> 0.068 ms
                         // (prologue + epilogue), 8 -> 0.483 (the loop is 0.2 ms), 1 -> 0.257, 3 -> 0.224, 2/4/5/6 -> 0.27-0.28:
                         // no single resource binds; the phases of a stage (weight loads, 22 fragment reads, 36 MFMAs, LDS
                         // writes, barrier) run back to back in 2 waves per SIMD that move in lockstep.

// Channel groups of 64 (2 K-chunks of 32); an odd chunk count (160, 288, 480, 928 channels) leaves the last group with one
template <int MT, int PRO, int EPI>
__global__ __launch_bounds__(S3_THREADS, 2 * S3_NW / 4 >= 2 ? 2 : 2) void conv_cl16_s3_kernel(const unsigned short* __restrict__ x,
                                                                     const unsigned short* __restrict__ wl,
                                                                     unsigned short* __restrict__ y,
                                                                     const float* __restrict__ in_ss,
                                                                     const float* __restrict__ scale_shift,
                                                                     const unsigned short* __restrict__ res, int relu,
                                                                     float* __restrict__ stat_sum,
                                                                     float* __restrict__ stat_sq, ClBnr bn, ClConv g, int prow) {
  constexpr int BM = MT * 16, KCG = 2;
  constexpr int ROWB = KCG * 64;                        // bytes per patch row: the group's channels
  constexpr int SLOTS = KCG * 4;                        // 16-byte slots per patch row
  constexpr int AB = BM * ROWB;                         // one weight stage: KCG blocks of [BM][64 B]
  constexpr int APIECES = BM * SLOTS, AIT = (APIECES + S3_THREADS - 1) / S3_THREADS;
  constexpr int OROW = BM * 2 + 16;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  const int PB = prow * ROWB;
  unsigned char* const patch0 = lds_raw;
  unsigned char* const ast0 = lds_raw + PB;
  float* const pro = (float*)(lds_raw + PB + 2 * AB);            // PRO 1: [2][Cin_p]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int W = g.Wi, H = g.Hi;
  const unsigned P = (unsigned)g.N * g.Ti * g.Hi * g.Wi;         // lattice == input positions == output positions
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, (int)(P * (unsigned)g.Cin_p * 2u), 0x00020000);
  unsigned bx = blockIdx.x, by = blockIdx.y;
  if (XCD_REMAP) {
    const unsigned nb = gridDim.x * gridDim.y, lin = blockIdx.x + blockIdx.y * gridDim.x, q8 = nb >> 3, r8 = nb & 7,
                   xcd = lin & 7, loc = lin >> 3;
    const unsigned unit = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + loc;
    by = unit / gridDim.x;
    bx = unit - by * gridDim.x;
  }
  const int m0 = by * BM;
  const unsigned p0 = bx * S3_BN;
  const int kcs = g.Cin_p >> 5, groups = (kcs + KCG - 1) / KCG;
  const int nst = SLV_S3_ABL == 7 ? 0 : groups * g.ntaps;      // ablation 7: prologue + epilogue only
  // ---- loaders
  u32x4 rp[S3_PIT];
  unsigned pvalid = 0;                               // bit i: patch piece i of this thread lies inside the tensor
  // (the thread index goes through an empty asm in both patch functions: their per-piece offsets, row / slot numbers and
  //  predicates are loop invariants that the compiler would otherwise keep in ~40 registers across the K loop)
  auto load_patch = [&](int cg) __attribute__((always_inline)) {
    pvalid = 0;
    int tl = tid;
    asm volatile("" : "+v"(tl));
#pragma unroll
    for (int i = 0; i < S3_PIT; ++i) {
      const int idx = tl + S3_THREADS * i, r = idx / SLOTS, s = idx - r * SLOTS;
      const int q = (int)p0 - W - 1 + r;               // 32-bit on purpose (P * Cin_p * 2 < 2^32: the host checks the tensor size)
      const int c = cg * (KCG * 32) + s * 8;
      const bool ok = r < prow && (unsigned)q < P && c < g.Cin_p;
      pvalid |= (unsigned)ok << i;
      const unsigned off = (unsigned)q * (unsigned)(g.Cin_p * 2) + (unsigned)c * 2u;
      rp[i] = SLV_S3_ABL == 6 ? (u32x4){off, 0u, 0u, 0u}
                              : __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, ok ? off : 0xFFFFFFF0u, 0, 0));
    }
  };
  auto store_patch = [&](int cg) __attribute__((always_inline)) {
    unsigned char* dst = patch0;
    int tl = tid;
    asm volatile("" : "+v"(tl));
#pragma unroll
    for (int i = 0; i < S3_PIT; ++i) {
      const int idx = tl + S3_THREADS * i, r = idx / SLOTS, s = idx - r * SLOTS;
      if (r < prow) {
        u32x4 v = rp[i];
        if constexpr (PRO == 1) {
          float sc[8], sh[8];
          const float* sp = pro + cg * (KCG * 32) + s * 8;
          *(f32x4*)sc = *(const f32x4*)sp;
          *(f32x4*)(sc + 4) = *(const f32x4*)(sp + 4);
          *(f32x4*)sh = *(const f32x4*)(sp + g.Cin_p);
          *(f32x4*)(sh + 4) = *(const f32x4*)(sp + g.Cin_p + 4);
          const u32x4 t = affine_relu8(v, sc, sh);
          v = ((pvalid >> i) & 1) ? t : (u32x4){0u, 0u, 0u, 0u};
        }
        *(u32x4*)(dst + r * ROWB + ((s ^ (r & (SLOTS - 1))) << 4)) = v;
      }
    }
  };
  // Weights go memory -> LDS directly (buffer_load_dwordx4 ... lds: no staging registers, no ds_write).  A stage's weight
  // block is [K-chunk of the group][BM rows][64 bytes] in LDS: linear in the piece index pc = tid + 256 i, and a wave's
  // 64 lanes of one instruction land in 1 KiB at M0 = block + i * 4096 + wave * 1024.  The swizzle moves to the source
  // side: LDS position (row, slot s') receives the memory piece (row, s' ^ swz(row)), and swz(row) is the same for every
  // i (BM % 16 == 0, 256 / 4 % 16 == 0).  Memory is [K-chunk][Mrows][64] (rows m0..m0+BM): ONE per-thread byte offset
  // plus, in the SGPR operand, what the stage adds (slab, group, i * 4096, the chunk's row gap).
  const __amdgpu_buffer_rsrc_t rwl = __builtin_amdgcn_make_buffer_rsrc((void*)wl, 0, (int)0x7FFFFFF0, 0x00020000);
  constexpr int KPC = BM * 4;                                           // pieces per K-chunk
  const unsigned kgap = (unsigned)((g.Mrows - BM) * 64);                // memory bytes between the chunks' row blocks
  const unsigned vo0 = (unsigned)((tid >> 2) * 64 + (((tid & 3) ^ cl_swz(tid >> 2)) << 4) + m0 * 64);
  constexpr int ISTR = (KPC % S3_THREADS) ? KPC / S3_THREADS : -1;      // the i whose pieces straddle the chunks
  const unsigned vos = vo0 + ((ISTR >= 0 && tid + S3_THREADS * ISTR >= KPC) ? kgap : 0u);
  const unsigned slab_bytes = (unsigned)(kcs * g.Mrows * 64), group_bytes = (unsigned)(KCG * g.Mrows * 64);
  typedef __attribute__((address_space(3))) void* lds_void;
  auto load_a = [&](int cg, int t, int buf) __attribute__((always_inline)) {
    const unsigned soff = (unsigned)t * slab_bytes + (unsigned)cg * group_bytes;      // taps are slabs 0..8 in order
    const int live = min(KCG, kcs - cg * KCG);
    unsigned char* dst = ast0 + buf * AB + wave * 1024;
#pragma unroll
    for (int i = 0; i < AIT; ++i) {
      const int lo = S3_THREADS * i, hi = lo + S3_THREADS - 1;
      const bool strad = i == ISTR;
      const int kl = strad ? (tid + lo >= KPC) : lo / KPC;              // K-chunk of the piece
      const bool ok = kl < live && SLV_S3_ABL != 3;                     // a chunk past the tensor's channels: zeros
      const unsigned so = soff + (unsigned)(i * (S3_THREADS * 16)) + ((!strad && lo >= KPC) ? kgap : 0u);
      if (hi < APIECES || tid + lo < APIECES)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rwl, (lds_void)(dst + i * (S3_THREADS * 16)), 16, ok ? (strad ? vos : vo0) : 0xFFFFFFF0u, so, 0, 0);
    }
  };
  f32x4 acc[MT][2];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#ifdef SLV_S3_TRACE      // timeline of wave 0 of 64 tiles (SLV_S3_TRACE_T0 ..): s_memtime at 5 points per stage + 4 per tile -> behind stat_sq's [Cout][tiles] floats (the caller allocates 32 KB more)
#ifndef SLV_S3_TRACE_T0
#define SLV_S3_TRACE_T0 0
#endif
#define S3_T(slot) if (trace) trace[(st * 5 + (slot))] = __builtin_amdgcn_s_memtime()
#define S3_TK(slot) if (trace) trace[(slot)] = __builtin_amdgcn_s_memtime()
  unsigned long long* trace = (EPI == 1 && tid == 0 && bx >= SLV_S3_TRACE_T0 && bx < SLV_S3_TRACE_T0 + 64) ? (unsigned long long*)(stat_sq + (size_t)g.Cout * gridDim.x) + (size_t)(bx - SLV_S3_TRACE_T0) * 64 : nullptr;   // BEHIND the statistics
#else
#define S3_T(slot)
#define S3_TK(slot)
#endif
  S3_TK(60);
  // ---- prologue of the pipeline
  if (nst > 0) {                                      // first patch and weight stage: in flight during the setup below
    load_patch(0);
    load_a(0, 0, 0);
  }
  if constexpr (PRO == 1) {
    for (int i = tid; i < 2 * g.Cin_p; i += S3_THREADS) {
      const int c = i % g.Cin_p, which = i / g.Cin_p;
      pro[i] = c < g.Cin ? in_ss[which * g.Cin + c] : 0.f;
    }
  }
  // ---- fragment lanes: positions of this lane's two 16-position tiles, their tap-validity masks
  const int fr = lane & 15, fk = lane >> 4;
  const int fsw = (fk ^ cl_swz(fr)) << 4;
  int prow_l[2];                                     // patch row of the position itself (tap offset 0)
  unsigned okm[2];                                   // bit (eh+1)*3 + (ew+1): the tap at offset (eh, ew) is inside the image
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int pl = wave * 32 + j * 16 + fr;
    const unsigned p = p0 + pl;
    prow_l[j] = pl + W + 1;
    const unsigned q = p / (unsigned)W;
    const int w = p - q * W, h = q % (unsigned)H;
    // rows of the 3 x 3 mask that exist (eh = -1, 0, 1 -> bits 0-2, 3-5, 6-8) AND its columns that exist (ew -> bit 0, 1, 2 of each row)
    const unsigned rows = (h > 0 ? 0x007u : 0u) | 0x038u | (h < H - 1 ? 0x1C0u : 0u);
    const unsigned cols = (w > 0 ? 0x049u : 0u) | 0x092u | (w < W - 1 ? 0x124u : 0u);
    okm[j] = p < P ? rows & cols : 0u;
  }
  if (nst > 0) {
    if constexpr (PRO == 1) __syncthreads();
    store_patch(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
  S3_TK(61);
  int cg = 0, t = 0, th = 0, tw = 0;                 // stage being COMPUTED; (th, tw) = (t / 3, t % 3)
  const int tsgn = ((((g.tap[0] >> 4) & 15) - 8 + g.boh) < 0) ? 1 : -1;      // forward: offsets -1..1 ascending; backward data: descending
  for (int rep = 0; rep < (SLV_S3_ABL == 8 ? 2 : 1); ++rep)      // ablation 8: the K loop twice
  for (int st = 0; st < nst; ++st) {
    if (SLV_S3_ABL == 8 && st == 0) { cg = 0; t = 0; th = 0; tw = 0; }
    int ncg = cg, nt = t + 1;                        // the next stage
    if (nt == g.ntaps) {
      nt = 0;
      ++ncg;
    }
    const bool more = st + 1 < nst;
    S3_T(0);
    if (more) load_a(ncg, nt, (st + 1) & 1);
    if (t == 0 && cg + 1 < groups) load_patch(cg + 1);
    S3_T(1);
    {
      const int eh = tsgn * (th - 1), ew = tsgn * (tw - 1);       // the 3 x 3 taps in slab order (s3_eligible checks it)
      const int roff = eh * W + ew;
      const unsigned bit = 1u << ((eh + 1) * 3 + ew + 1);
      const unsigned char* A = ast0 + (st & 1) * AB;
      const unsigned char* Bp = patch0;
      const int live = min(KCG, kcs - cg * KCG);        // K-chunks of this group that exist
      static_assert(KCG == 2, "the stage body below is written for two K-chunks per stage");
      // Order inside a stage (fixed with sched_barrier: left alone, the scheduler re-uses ONE weight-fragment register
      // and waits a full LDS round trip before every MFMA pair): all fragment reads of chunk 0 and the position
      // fragments of chunk 1, then chunk 0's MFMA pairs, each followed by the read that refills its weight register
      // with chunk 1's fragment, then the next stage's weights go to LDS (their global loads were issued at the top
      // of the stage), then chunk 1's MFMAs with nothing left to wait for.
      auto rdb = [&](int kl, int j) __attribute__((always_inline)) {
        const int r = prow_l[j] + roff;
        return SLV_S3_ABL == 2 ? (u32x4){(unsigned)r, 1u, 2u, 3u}
                               : *(const u32x4*)(Bp + r * ROWB + (((kl * 4 + fk) ^ (r & (SLOTS - 1))) << 4));
      };
      auto rda = [&](int kl, int i) __attribute__((always_inline)) {
        return SLV_S3_ABL == 2 ? __builtin_bit_cast(bf16x8, (u32x4){(unsigned)(i + roff), 5u, 6u, 7u})
                               : *(const bf16x8*)(A + kl * (BM * 64) + (i * 16 + fr) * 64 + fsw);
      };
      auto mma = [&](int i, const bf16x8& av, const bf16x8* bv) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          if (SLV_S3_ABL == 1) acc[i][j][0] += __builtin_bit_cast(float, __builtin_bit_cast(u32x4, av)[0] ^ __builtin_bit_cast(u32x4, bv[j])[0]);
          else acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, bv[j], acc[i][j], 0, 0, 0);
        }
      };
      u32x4 bv0[2], bv1[2];
      bf16x8 a[MT], b[2];
#pragma unroll
      for (int j = 0; j < 2; ++j) bv0[j] = rdb(0, j);
#pragma unroll
      for (int i = 0; i < MT; ++i) a[i] = rda(0, i);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < 2; ++j) b[j] = __builtin_bit_cast(bf16x8, (okm[j] & bit) ? bv0[j] : (u32x4){0u, 0u, 0u, 0u});
      // ONE copy of chunk 0's MFMAs for both cases (a second copy under `live == 1` made the register allocator rename
      // the accumulators per path and reconcile them with 72 v_mov per stage at the loop's back edge: a quarter of
      // the kernel's VALU instructions, on a kernel whose SIMDs spend 52 % of their cycles issuing VALU)
      const bool two = live > 1;
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        mma(i, a[i], b);
        if (two) {
          if (i == MT - 1) {
#pragma unroll
            for (int j = 0; j < 2; ++j) bv1[j] = rdb(1, j);
          }
          a[i] = rda(1, i);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      S3_T(2);
      if (two) {
#pragma unroll
        for (int j = 0; j < 2; ++j) b[j] = __builtin_bit_cast(bf16x8, (okm[j] & bit) ? bv1[j] : (u32x4){0u, 0u, 0u, 0u});
#pragma unroll
        for (int i = 0; i < MT; ++i) mma(i, a[i], b);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    S3_T(3);
    if (t == g.ntaps - 1 && cg + 1 < groups) {       // ONE patch buffer: every wave is done with this group's rows first
      __syncthreads();
      store_patch(cg + 1);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the next stage's weights have landed in LDS
    if (SLV_S3_ABL != 5) __syncthreads();
    S3_T(4);
    cg = ncg;
    t = nt;
    if (++tw == 3) {
      tw = 0;
      if (++th == 3) th = 0;
    }
  }
  S3_TK(62);
  // ---- epilogue (cf. conv_cl16.hip): transposed tile [position][cout] through LDS, 16-byte stores along the channels
  __syncthreads();
  unsigned char* ot = lds_raw;                          // [S3_BN][OROW]
  float* ssl = (float*)(lds_raw + S3_BN * OROW);        // EPI 0: [2][BM]; EPI 1: [waves][2][BM]; EPI 2: [4][BM] + [waves][2][BM]
  float* red = ssl + (EPI == 2 ? 4 * BM : 0);
  if (EPI == 0 && scale_shift) {
    for (int i = tid; i < 2 * BM; i += S3_THREADS) {
      const int c = m0 + (i % BM);
      ssl[i] = c < g.Cout ? scale_shift[(i / BM) * g.Cout + c] : 0.f;
    }
    __syncthreads();
  }
  if constexpr (EPI == 2) {
    for (int i = tid; i < 4 * BM; i += S3_THREADS) {
      const int c = m0 + (i % BM), which = i / BM;
      ssl[i] = c < g.Cout ? (which < 2 ? bn.ss[which * g.Cout + c] : bn.mi[(which - 2) * g.Cout + c]) : 0.f;
    }
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    const int co = m0 + i * 16 + fk * 4;
    f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
    if (EPI == 0 && scale_shift) {
      sc = *(const f32x4*)(ssl + i * 16 + fk * 4);
      sh = *(const f32x4*)(ssl + BM + i * 16 + fk * 4);
    }
    float ps[4] = {0.f, 0.f, 0.f, 0.f}, pq[4] = {0.f, 0.f, 0.f, 0.f};
    f32x4 bs = {0.f, 0.f, 0.f, 0.f}, bh = bs, bmean = bs, binv = bs;
    if constexpr (EPI == 2) {
      bs = *(const f32x4*)(ssl + i * 16 + fk * 4);
      bh = *(const f32x4*)(ssl + BM + i * 16 + fk * 4);
      bmean = *(const f32x4*)(ssl + 2 * BM + i * 16 + fk * 4);
      binv = *(const f32x4*)(ssl + 3 * BM + i * 16 + fk * 4);
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int pl = wave * 32 + j * 16 + fr;
      const unsigned p = p0 + pl;
      float v[4];
      if constexpr (EPI != 1) {
        uint2 rr = make_uint2(0u, 0u);
        if (res && p < P && co < g.Cout_p) rr = *(const uint2*)(res + (size_t)p * g.Cout_p + co);
        if (!scale_shift && !relu) {      // backward data: the accumulator (+ addend) as it is (cf. conv_cl16.hip)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            v[r] = acc[i][j][r];
            if (res) v[r] += bf2f((unsigned short)((r < 2 ? rr.x : rr.y) >> ((r & 1) * 16)));
          }
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float tt = __builtin_fmaf(acc[i][j][r], sc[r], sh[r]);     // (scale, shift are zero for channels >= Cout, the residual's padding is zero: no select)
            if (res) tt += bf2f((unsigned short)((r < 2 ? rr.x : rr.y) >> ((r & 1) * 16)));
            if (relu) tt = fmaxf(tt, 0.f);
            v[r] = tt;
          }
        }
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = acc[i][j][r];
      }
      const unsigned lo = pack_bf2(v[0], v[1]), hi = pack_bf2(v[2], v[3]);
      *(uint2*)(ot + pl * OROW + (i * 16 + fk * 4) * 2) = make_uint2(lo, hi);
      if constexpr (EPI == 2) {                          // BatchNorm-backward sums of the source layer (ClBnr)
        uint2 xr = make_uint2(0u, 0u);
        if ((p < P) && co < g.Cout_p) xr = *(const uint2*)(bn.x + (size_t)p * g.Cout_p + co);
        const float gg[4] = {bf_lo(lo), bf_hi(lo), bf_lo(hi), bf_hi(hi)};
        const float xx[4] = {bf_lo(xr.x), bf_hi(xr.x), bf_lo(xr.y), bf_hi(xr.y)};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float gm = bn_affine(xx[r], bs[r], bh[r]) > 0.f ? gg[r] : 0.f;
          ps[r] += gm;
          pq[r] += gm * ((xx[r] - bmean[r]) * binv[r]);
        }
      }
    }
    if constexpr (EPI == 2) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float a = row16_sum(ps[r]), b = row16_sum(pq[r]);
        if (fr == 0) {
          red[(wave * 2 + 0) * BM + i * 16 + fk * 4 + r] = a;
          red[(wave * 2 + 1) * BM + i * 16 + fk * 4 + r] = b;
        }
      }
    }
  }
  if constexpr (EPI == 1)                                // statistics of this wave's 32 rows, on the matrix cores
    wave_tile_stats<MT>(ot + wave * 32 * OROW, OROW, lane, red + (wave * 2 + 0) * BM, red + (wave * 2 + 1) * BM);
  __syncthreads();
  if constexpr (EPI >= 1) {                              // the waves' partials in fixed order
    for (int i = tid; i < 2 * BM; i += S3_THREADS) {
      const int c = i % BM, which = i / BM;
      if (m0 + c < g.Cout) {
        float tt = 0.f;
#pragma unroll
        for (int wv = 0; wv < S3_NW; ++wv) tt += red[(wv * 2 + which) * BM + c];
        if constexpr (EPI == 1) (which ? stat_sq : stat_sum)[(size_t)(m0 + c) * gridDim.x + bx] = tt;
        else bn.part[((size_t)(m0 + c) * bn.nslots + bn.slot0 + bx) * 2 + which] = tt;
      }
    }
  }
  const int c_lo = m0, c_hi = min(m0 + BM, g.Cout_p);
  const int c_end = (by == gridDim.y - 1) ? g.Cout_p : c_hi;
  const int pieces = (c_end - c_lo) >> 3;
  // a thread keeps its 16-byte column piece and walks down the rows: rpp = threads / pieces rows per pass (one division
  // per thread instead of index arithmetic per store; P * Cout_p * 2 < 2^32, so 32-bit byte offsets)
  const int rpp = S3_THREADS / pieces, pl0 = tid / pieces, pc = tid - pl0 * pieces;
  if (pl0 < rpp) {
    const bool cval = c_lo + pc * 8 < c_hi;
    const unsigned char* src = ot + pl0 * OROW + pc * 16;
    unsigned char* yb = (unsigned char*)y;
    unsigned off = ((p0 + (unsigned)pl0) * (unsigned)g.Cout_p + (unsigned)(c_lo + pc * 8)) * 2u;
    const unsigned dstep = (unsigned)(rpp * g.Cout_p * 2);
    for (int pl = pl0; pl < S3_BN && p0 + (unsigned)pl < P; pl += rpp, src += rpp * OROW, off += dstep) {
      u32x4 val = {0u, 0u, 0u, 0u};
      if (cval) val = *(const u32x4*)src;
      if (SLV_S3_ABL != 4 || val[0] == 0x12345678u) *(u32x4*)(yb + off) = val;
    }
  }
#ifdef SLV_S3_TRACE
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  S3_TK(63);
#endif
}

static size_t s3_lds_bytes(int mt, int prow, int cin_p, bool pro) {
  const size_t bm = (size_t)mt * 16, kcg = 2;
  const size_t k_loop = (size_t)prow * kcg * 64 + 2 * bm * kcg * 64 + (pro ? 2 * (size_t)cin_p * 4 : 0);
  const size_t epi = (size_t)S3_BN * (bm * 2 + 16) + (size_t)(4 + 2 * S3_NW) * bm * 4;
  return k_loop > epi ? k_loop : epi;
}

// Does this launch fit the patch kernel?  Stride 1, lattice == input == output positions, every tap within one row /
// column of the position, the patch (256 + 2W + 2 rows) within the loader's reach and the LDS.
static bool s3_eligible(const ClConv& g) {
  if (g.ntaps != 9) return false;
  if (g.Lt != g.Ti || g.Lh != g.Hi || g.Lw != g.Wi || g.To != g.Ti || g.Ho != g.Hi || g.Wo != g.Wi) return false;
  if (g.bmt != 1 || g.bmh != 1 || g.bmw != 1 || g.omt != 1 || g.omh != 1 || g.omw != 1 || g.oot || g.ooh || g.oow) return false;
  const int sgn = ((((g.tap[0] >> 4) & 15) - 8 + g.boh) < 0) ? 1 : -1;
  for (int t = 0; t < 9; ++t) {        // tap t = weight slab t at offset sgn * (t / 3 - 1, t % 3 - 1): what the kernel computes
    const int dt = (g.tap[t] & 15) - 8 + g.bot, dh = ((g.tap[t] >> 4) & 15) - 8 + g.boh, dw = ((g.tap[t] >> 8) & 15) - 8 + g.bow;
    if (dt != 0 || dh != sgn * (t / 3 - 1) || dw != sgn * (t % 3 - 1) || (g.tap[t] >> 12) != t) return false;
  }
  const int prow = S3_BN + 2 * g.Wi + 2;
  if (prow * 8 > S3_PIT * S3_THREADS) return false;
  // the LDS budget belongs to the eligibility (the same answer at plan time -- slv_cl16_conv_nblk -- and at launch, for
  // every tile height and with the prologue table): a launch that does not fit goes to the general kernel
  if (s3_lds_bytes(9, prow, g.Cin_p, true) > 160 * 1024) return false;
  return true;
}

template <int MT, int PRO, int EPI>
static int s3_launch_one(const ClConv& g, const void* x, const void* wl, void* y, const float* in_ss,
                         const float* scale_shift, const void* res, int relu, float* stat_sum, float* stat_sq,
                         const ClBnr& bnr, hipStream_t st) {
  constexpr int BM = MT * 16;
  const int prow = S3_BN + 2 * g.Wi + 2;
  const size_t lds = s3_lds_bytes(MT, prow, g.Cin_p, PRO != 0);      // <= 160 KB: s3_eligible checked the largest variant
  static bool attr_set = false;                          // per instantiation; idempotent, so a race is harmless
  if (!attr_set) {
    SLV_HIP(hipFuncSetAttribute((const void*)conv_cl16_s3_kernel<MT, PRO, EPI>,
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_set = true;
  }
  const unsigned P = (unsigned)g.N * g.Ti * g.Hi * g.Wi;
  dim3 grid((P + S3_BN - 1) / S3_BN, g.Mrows / BM);
  hipLaunchKernelGGL((conv_cl16_s3_kernel<MT, PRO, EPI>), grid, dim3(S3_THREADS), lds, st, (const unsigned short*)x,
                     (const unsigned short*)wl, (unsigned short*)y, in_ss, scale_shift, (const unsigned short*)res, relu,
                     stat_sum, stat_sq, bnr, g, prow);
  return 0;
}

static bool s3_enabled() {
  static const bool enabled = []() {
    const char* e = getenv("SELAVI_CL16_S3");
    return !(e && e[0] == '0');
  }();
  return enabled;
}
bool cl16_s3_applies(const ClConv& g) { return s3_enabled() && s3_eligible(g); }
int cl16_s3_positions() { return S3_BN; }

// returns 1 when the launch was taken by the patch kernel, 0 when it does not apply, < 0 on error
int cl16_s3_try(const ClConv& g, int mt, const void* x, const void* wl, void* y, const float* in_ss,
                const float* scale_shift, const void* res, int relu, float* stat_sum, float* stat_sq, const ClBnr& bnr,
                hipStream_t st) {
  if (!cl16_s3_applies(g)) return 0;
  const int pro = in_ss ? 1 : 0, epi = bnr.part ? 2 : (stat_sum ? 1 : 0);
  int rc = 0;
#define SLV_S3_K(MT_, PRO_, EPI_) \
  rc = s3_launch_one<MT_, PRO_, EPI_>(g, x, wl, y, in_ss, scale_shift, res, relu, stat_sum, stat_sq, bnr, st)
#define SLV_S3_MT(MT_)                                  \
  do {                                                  \
    if (epi == 2) SLV_S3_K(MT_, 0, 2);                  \
    else if (pro == 0 && epi == 0) SLV_S3_K(MT_, 0, 0); \
    else if (pro == 1 && epi == 0) SLV_S3_K(MT_, 1, 0); \
    else if (pro == 0 && epi == 1) SLV_S3_K(MT_, 0, 1); \
    else SLV_S3_K(MT_, 1, 1);                           \
  } while (0)
  if (mt == 4) SLV_S3_MT(4);
  else if (mt == 8) SLV_S3_MT(8);
  else SLV_S3_MT(9);
#undef SLV_S3_MT
#undef SLV_S3_K
  if (rc) return rc;
  rc = launch_check("slv_cl16_conv");
  return rc ? rc : 1;
}

}  // namespace slv
