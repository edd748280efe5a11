// Implicit-GEMM convolution / GEMM core for gfx950 (MI355X), fp32 in / fp32 accumulate on the
// matrix cores (v_mfma_f32_16x16x4_f32: exact fp32, 157 TF peak = fp32 vector peak).
//
// One kernel template covers the GEMM-shaped ops of the SeLaVi training step
// (reference: torchvision Conv3d/Conv2d + autograd called from model.py:145-166, main.py:284-301):
//   MODE_CONV   Y[m, q]   = sum_{c,tap} A[m, (c,tap)] * pro(X)[c, q*mul + delta(tap)]
//               - forward conv:   m = co, lattice q = output positions, mul = stride, delta = tap - pad
//               - backward data:  m = ci, one launch per stride-parity class of the INPUT positions
//                                 (q*stride + class), taps restricted to the class, delta = (class+pad-tap)/stride
//                                 -> no wasted MFMA work for stride-2 layers, all offsets stay linear
//   MODE_WGRAD  dW[co, (ci,tap)] = sum_p dXout[co, p] * pro(X)[ci, p*stride + tap - pad]
//   MODE_GEMM   C[m, n]   = sum_k A[m, k] * B[n, k]          (dense "NT" GEMM for the heads)
// with the BatchNorm that surrounds every conv of the model fused into the operand loaders:
//   PRO_ACT  v = relu?(x*scale[c] + shift[c])                (consumer-side BN apply + ReLU)
// so activated tensors are never materialised in HBM (the BN-backward gradient is materialised once
// per layer by slv_bn_bwd_apply: measured 1.8x faster than folding it into the loaders), and the forward
// epilogue emits per-channel sum / sum-of-squares partials (training-mode batch statistics); the
// backward-data epilogue can emit the BatchNorm-backward partial sums of the producing layer (EPI_BNR).
// K of MODE_CONV is ordered channel-major (per-k table, the stems) or tap-major in 16-channel groups
// (KORD_TAP: one tap per 16-deep chunk -> validity/address math per chunk, SGPR channel offsets).
// Deep-K launches with few columns are split over K (deterministic partials + fixed-order reduce).
//
// Tiling (wave64, 256 threads = 4 waves): block tile (MT*16) x (NT*64) x 16; each wave owns all
// MT*16 rows and NT*16 of the columns, i.e. MT x NT accumulators of 16x16 (4 VGPRs each).  The M
// tile is chosen per layer from {64,128,144,240} so the odd channel counts of R(2+1)D-18
// (45/144/230/460/921) waste < 5 % of the MFMA work.  LDS tiles are [rows][16+2] for K-contiguous
// operands and [16][BN+16] for position-contiguous operands; both give conflict-free ds_read_b32
// fragment reads (bank = (a/4) % 32 inside each 32-lane half).
//
// Loads: raw buffer loads (SGPR descriptor + 32-bit byte offset).  Out-of-range elements (padding,
// tile tails) are given an out-of-bounds offset and come back as 0 from the hardware range check,
// so the loaders have no branches and no 64-bit address math; tap validity is a per-thread bitmask
// computed once per block.  Global->LDS is register staged and double buffered: loads of chunk c+1
// are issued before the MFMAs of chunk c, masking/prologue math runs when the registers are written
// to LDS after those MFMAs.  One barrier per chunk.
#pragma once
#include <type_traits>

#include "common.hpp"

namespace slv {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

enum { MODE_CONV = 0, MODE_WGRAD = 2, MODE_GEMM = 3 };
enum { PRO_NONE = 0, PRO_ACT = 1 };
// K ordering of MODE_CONV: channel-major (k = c*ntaps + tap, the nn.Conv weight layout, per-k table) or
// tap-major (k = ((c/16)*ntaps + tap)*16 + c%16 over Cpad = round16(C) channels, transformed weights): every
// 16-deep chunk then has ONE tap, so padding validity / address math is per chunk instead of per element,
// and the taps of a 16-channel group are consecutive chunks (L1/L2 reuse of the shifted input rows).
enum { KORD_CHAN = 0, KORD_TAP = 1 };
// CONV epilogue: plain store (+ addend, + forward BN statistics) or store + BatchNorm-backward partial sums
// (IgemmArgs::R).  A template parameter: the plain kernels keep their register budget.
enum { EPI_PLAIN = 0, EPI_BNR = 1, EPI_EVAL = 2 };     // EPI_EVAL: igemm3.hpp only (bias + addend + ReLU, eval-mode forward)

constexpr unsigned OOB = 0xFFFFFFF0u;  // byte offset beyond any buffer -> load returns 0

struct IgemmArgs {
  // operands (device pointers + byte sizes for the buffer descriptors)
  const float* A;  unsigned A_bytes;    // CONV/GEMM: dense [M][Kd]; WGRAD: gradient tensor (conv-output side)
  const float* B;  unsigned B_bytes;    // CONV: gathered tensor; WGRAD: conv input; GEMM: dense [N][Kd]
  const float* pb;   // B prologue params (PRO_ACT): [2][Cb] = scale, shift
  const int2* tab;   // per-k (CONV) / per-column (WGRAD) entries {element offset, tap | chan << 8}; padded with
                     // invalid entries {0, 63} to a multiple of 16 (+16).
                     // CONV with KORD_TAP: one entry per 16-deep CHUNK {element offset of (tap, first channel),
                     // tap | first channel << 8}
  int kord;          // CONV: KORD_CHAN / KORD_TAP
  int vec_b;         // WGRAD: 16-byte loads of the gathered operand: 0 no, 1 / 2 = template flag VB
  unsigned sprod4;   // KORD_TAP: bytes between consecutive channels of the gathered tensor (S0*S1*S2*4)
  const int* tapd;   // per-tap packed deltas: (d0+64) | (d1+64)<<8 | (d2+64)<<16, 64 entries
  float* C;          // output
  const float* E;    // optional epilogue addend, same indexing as C
  const float* bias; // GEMM: optional per-column bias; CONV with EPI_EVAL: per-ROW (output channel) bias [M]
  int epi_relu;      // EPI_EVAL: ReLU behind bias + addend
  float* stat_sum;   // CONV: [M][nblkN] per-channel partial sums of the output (or null)
  float* stat_sq;
  // CONV (backward-data) epilogue: BatchNorm-backward reductions of the layer that PRODUCED this launch's
  // input tensor: with g = the value stored at C (incl. addend), x = R at the same index,
  // g' = g * (x*s + h > 0):  rpart[m][rslot0 + nblk] = { sum g', sum g' * (x - mean) * invstd }
  const float* R;    // raw conv output of that layer (same shape/indexing as C) or null
  const float* rss;  // [2][M] scale, shift (ReLU mask)
  const float* rmi;  // [2][M] mean, invstd
  float* rpart;      // [M][rslots][2]
  int rslots, rslot0;
  int M, Kd;
  long long Ntot;    // columns: lattice positions (CONV) or Cin*taps (WGRAD) or N (GEMM)
  int nblkM, nblkN;
  int Cb;            // channels of the B tensor (prologue param stride)
  // gather lattice (CONV): column n -> (b, q0, q1, q2); source coord = q*mul + delta(tap), bounds S
  int Q0, Q1, Q2, mul0, mul1, mul2, S0, S1, S2;
  long long sbatch;  // elements per sample of the B tensor
  // destination lattice (CONV): dst coord = q*dmul + dorg inside [D0][D1][D2]
  int dmul0, dmul1, dmul2, dorg0, dorg1, dorg2, D0, D1, D2;
  int ntaps;
  int b_pro, b_relu;
  // WGRAD geometry
  int Cin, Ti, Hi, Wi, Cout, To, Ho, Wo, st, sh, sw, pt, ph, pw;
  int chunks_per_split;   // WGRAD: always; CONV: > 0 selects split-K (partials to C + split*split_stride)
  long long split_stride; // CONV split-K: elements between the partial outputs of consecutive K-slices
  long long Ptot;
  int ldc;
  FastDiv dPout, dHoWo, dWo;
};

__device__ __forceinline__ float apply_act(float x, float s, float h, int relu) {
  const float v = x * s + h;
  return relu ? fmaxf(v, 0.f) : v;
}
__device__ __forceinline__ float bload(__amdgpu_buffer_rsrc_t r, unsigned off) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, off, 0, 0));
}
__device__ __forceinline__ float bload_s(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {  // soff: SGPR
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
}
__device__ __forceinline__ f32x4 bload4(__amdgpu_buffer_rsrc_t r, unsigned off) {  // 16-byte aligned offset
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0));
}

// VA: the A operand is read with 16-byte loads (4 consecutive k per lane).  Host-checked
// preconditions: CONV/GEMM Kd % 4 == 0; WGRAD To*Ho*Wo % 4 == 0.
// PRO: prologue of the gathered B operand (PRO_NONE / PRO_ACT) -- a template parameter so that the
// steady-state loop stays one basic block.
// MF: MFMA shape.  0: v_mfma_f32_16x16x4_f32 (tile edge 16, MT x NT tiles of 16x16 per wave, block (MT*16) x (NT*64));
//     1: v_mfma_f32_32x32x2_f32 (tile edge 32, block (MT*32) x (NT*128)): one A + one B fragment read per 4096 FLOP
//        instead of per 2048 -- the LDS+MFMA-only loop reaches 145 TF with it against 127-135 TF
//        (tools/mfma_lds_bench.hip); usable where the row count pads well to 32/64/96/128.
// VB (MODE_WGRAD, with VA): the gathered operand is read with 16-byte loads as well.  Host-checked: spatial stride 1,
// "same" spatial padding with kh, kw in {1, 3} (Ho = Hi, Wo = Wi), Wo % 4 == 0, 16-byte aligned input.  A quad of
// consecutive output positions (one row, wo % 4 == 0) then reads the 4 consecutive input elements wo+dw .. wo+dw+3
// of row (to*st + dt, ho + dh): the row is valid or padded as a whole; for dw = -1 / +1 only the first / last
// element of a quad at the left / right image border is padding -- there the aligned quad is loaded and shifted.
// VB = 1: kh = kw = 1 (temporal / pointwise convs; only Ho*Wo % 4 == 0 needed), VB = 2: the general form.
template <int MODE, int MT, int NT, bool VA, int PRO, int KORD = KORD_CHAN, int EPI = EPI_PLAIN, int MF = 0, int VB = 0>
#ifndef SLV_LB_CONV
#define SLV_LB_CONV 3
#endif
#ifndef SLV_LB_WGRAD
#define SLV_LB_WGRAD 2
#endif
__global__ __launch_bounds__(256, ((MF == 0 && MT >= 15) ? 2 : (MODE == MODE_WGRAD ? SLV_LB_WGRAD : SLV_LB_CONV))) void igemm_kernel(const IgemmArgs g) {
  constexpr int RT = MF ? 32 : 16;   // MFMA tile edge
  constexpr int AR = MF ? 16 : 4;    // accumulator registers per MFMA tile
  constexpr int KPS = MF ? 2 : 4;    // k per MFMA step
  constexpr int BM = MT * RT, BN = NT * RT * 4;
  constexpr int MR16 = BM / 16;      // 16-row passes of the scalar A loader
  using Acc = std::conditional_t<MF != 0, f32x16, f32x4>;
  constexpr int AS = 18;
  constexpr bool BKF = (MODE == MODE_WGRAD || MODE == MODE_GEMM);  // B tile K-contiguous?
  constexpr int BS = BKF ? 18 : (BN + 16);
  constexpr int A_ELEMS = BM * AS;
  constexpr int B_ELEMS = BKF ? BN * 18 : 16 * BS;
  constexpr int BROWS = BN / 16;  // B staging registers per thread (both layouts)
  constexpr int P_ELEMS = (MODE == MODE_WGRAD && PRO == PRO_ACT) ? 2 * BN : 0;
  __shared__ float smem[2 * (A_ELEMS + B_ELEMS) + P_ELEMS];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

  // ---- XCD-aware block remap (bijective): the tiles that share operands share an XCD's L2
  int mblk, nblk, split;
  {
    const int nb = gridDim.x, id = blockIdx.x;
    const int q = nb >> 3, r = nb & 7, xcd = id & 7, loc = id >> 3;
    const int nid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
    const int per = g.nblkM * g.nblkN;
    split = nid / per;  // WGRAD: K-slice
    const int rem_ = nid - split * per;
    mblk = rem_ % g.nblkM;
    nblk = rem_ / g.nblkM;
  }
  const int m0 = mblk * BM;
  const long long n0 = (long long)nblk * BN;
  const int mrem = g.M - m0;  // valid rows in this block (may exceed BM)

  const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc((void*)g.A, 0, (int)g.A_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rB = __builtin_amdgcn_make_buffer_rsrc((void*)g.B, 0, (int)g.B_bytes, 0x00020000);

  // ---- per-thread constants of the loaders
  const int a_kk = tid & 15, a_r = tid >> 4;  // K-contiguous loader: 16 k's x 16 rows per pass

  // CONV: one lattice column per thread, BROWS k's per chunk
  int nl = 0, kg = 0;
  unsigned lbase = 0;           // element offset of this thread's column in the B tensor (may wrap; masked)
  unsigned mlo = 0, mhi = 0;    // tap validity bitmask of this column
  if constexpr (MODE == MODE_CONV) {
    nl = tid % BN;
    kg = __builtin_amdgcn_readfirstlane(tid / BN);
    const long long n = n0 + nl;
    const bool nvalid = n < g.Ntot;
    const long long nn = nvalid ? n : 0;
    const int NQ = g.Q0 * g.Q1 * g.Q2, Q12 = g.Q1 * g.Q2;
    const int b = (int)(nn / NQ);
    int rem = (int)(nn - (long long)b * NQ);
    const int q0 = rem / Q12;
    rem -= q0 * Q12;
    const int q1 = rem / g.Q2, q2 = rem - q1 * g.Q2;
    const int c0 = q0 * g.mul0, c1 = q1 * g.mul1, c2 = q2 * g.mul2;
    lbase = (unsigned)((long long)b * g.sbatch + (long long)c0 * (g.S1 * g.S2) + c1 * g.S2 + c2);
    if constexpr (KORD == KORD_TAP) lbase += (unsigned)(kg * BROWS) * (g.sprod4 >> 2);  // this thread's first channel row
    if (nvalid) {
      for (int t = 0; t < g.ntaps; ++t) {
        const int d = g.tapd[t];
        const bool ok = (unsigned)(c0 + (d & 255) - 64) < (unsigned)g.S0 &&
                        (unsigned)(c1 + ((d >> 8) & 255) - 64) < (unsigned)g.S1 &&
                        (unsigned)(c2 + ((d >> 16) & 255) - 64) < (unsigned)g.S2;
        if (t < 32) mlo |= (ok ? 1u : 0u) << t;
        else mhi |= (ok ? 1u : 0u) << (t - 32);
      }
    }
  }
  // WGRAD: this thread's reduction position p = chunk*16 + a_kk
  const int HWi = g.Hi * g.Wi, THWi = g.Ti * HWi;
  const int HoWo = g.Ho * g.Wo, Pout = g.To * HoWo;
  unsigned wp = 0;
  int nchunks;
  if constexpr (MODE == MODE_WGRAD) {
    const long long c0 = (long long)split * g.chunks_per_split;
    const long long call = (g.Ptot + 15) / 16;
    long long c1 = c0 + g.chunks_per_split;
    if (c1 > call) c1 = call;
    nchunks = (int)(c1 > c0 ? c1 - c0 : 0);
    wp = (unsigned)(c0 * 16 + a_kk);
  } else {
    nchunks = (g.Kd + 15) / 16;
  }
  int kbase = 0;  // CONV split-K: first k of this block's K-slice
  if constexpr (MODE == MODE_CONV) {
    if (g.chunks_per_split > 0) {
      const int c0 = split * g.chunks_per_split;
      int c1 = c0 + g.chunks_per_split;
      if (c1 > nchunks) c1 = nchunks;
      nchunks = c1 > c0 ? c1 - c0 : 0;
      kbase = c0 * 16;
    }
  }

  // staging registers hold RAW loaded values; masking + prologue math run in store_chunk
  float ra[VA ? 1 : MR16];
  constexpr int NP = (BM + 63) / 64;  // VA: passes of 64 rows x 4 k-quads
  f32x4 ra4[VA ? NP : 1];
  (void)ra; (void)ra4;
  const int v_kq = tid & 3, v_mr = tid >> 2;
  float rb[BROWS];
  unsigned okA = 0, okB = 0;  // validity bits (only consulted when a prologue must be masked)
  float sp0[BROWS], sp1[BROWS];  // CONV + PRO_ACT: per-k scale/shift (wave-uniform)
  (void)sp0; (void)sp1;
  (void)okA; (void)okB;

  // WGRAD: loop-invariant table entries of this thread's B rows and per-channel params in LDS
  int wt_off[BROWS], wt_d[BROWS];
  (void)wt_off; (void)wt_d;
  float* pBs = smem + 2 * (A_ELEMS + B_ELEMS);  // [2][BN]
  constexpr int NBQ = BN / 64;                   // VB: 64-column passes (column = tid >> 2 + 64 i, k quad = tid & 3)
  f32x4 rb4[VB ? NBQ : 1];
  int vb_off[NBQ], vb_dt[NBQ], vb_dh[NBQ], vb_dw[NBQ];
  float vb_s[NBQ], vb_h[NBQ];
  unsigned edgeLo = 0, edgeHi = 0;   // per column-pass bit: the quad was loaded one element to the right / left
  (void)rb4; (void)vb_off; (void)vb_dt; (void)vb_dh; (void)vb_dw; (void)vb_s; (void)vb_h; (void)edgeLo; (void)edgeHi;
  if constexpr (MODE == MODE_WGRAD && VB) {
#pragma unroll
    for (int i = 0; i < NBQ; ++i) {
      const long long n = n0 + (tid >> 2) + 64 * i;
      const int2 e = g.tab[n < g.Ntot ? n : g.Ntot];          // entry Ntot is an invalid pad entry (tap 63)
      vb_off[i] = e.x;
      const bool pad_entry = (e.y & 63) == 63;
      const int d = pad_entry ? 0 : g.tapd[e.y & 63];
      vb_dt[i] = pad_entry ? (1 << 20) : ((d & 255) - 64);            // deltas include "- pad"
      vb_dh[i] = ((d >> 8) & 255) - 64;
      vb_dw[i] = ((d >> 16) & 255) - 64;
      const int ch = ((e.y & 63) == 63) ? 0 : (e.y >> 8);
      vb_s[i] = (PRO == PRO_ACT) ? g.pb[ch] : 1.f;
      vb_h[i] = (PRO == PRO_ACT) ? g.pb[g.Cin + ch] : 0.f;
    }
  }
  if constexpr (MODE == MODE_WGRAD && !VB) {
#pragma unroll
    for (int i = 0; i < BROWS; ++i) {
      const long long n = n0 + a_r + 16 * i;
      const int2 e = g.tab[n < g.Ntot ? n : g.Ntot];  // entry Ntot is an invalid pad entry (tap 63)
      wt_off[i] = e.x;
      wt_d[i] = ((e.y & 63) == 63) ? -1 : g.tapd[e.y & 63];
    }
    if constexpr (PRO == PRO_ACT) {
      for (int i = tid; i < 2 * BN; i += 256) {
        const int which = i / BN, nn = i - which * BN;
        const long long n = n0 + nn;
        pBs[i] = (n < g.Ntot) ? g.pb[which * g.Cin + (g.tab[n].y >> 8)] : 0.f;
      }
      __syncthreads();  // store_chunk(0) of OTHER threads reads these before the first loop barrier
    }
  }

  // NB: every lambda is force-inlined -- an out-of-line lambda captures the register arrays by
  // reference and drags the accumulators into scratch memory.
  // ---------------------------------------------------------------- global -> registers (raw, branch-free)
  auto load_chunk = [&](int c) __attribute__((always_inline)) {
    const int k0 = kbase + c * 16;
    if constexpr (MODE != MODE_WGRAD) {
      // A: dense [M][Kd]; rows >= M fall outside the buffer (-> 0), the k tail is neutralised by B == 0
      if constexpr (VA) {
        const unsigned abase = (unsigned)(((m0 + v_mr) * (long long)g.Kd + k0 + 4 * v_kq) * 4);
        const bool kok = (MODE != MODE_GEMM) || (k0 + 4 * v_kq < g.Kd);
#pragma unroll
        for (int i = 0; i < NP; ++i) {
          const unsigned off = abase + (unsigned)(i * 64 * g.Kd * 4);
          ra4[i] = bload4(rA, (kok && v_mr + 64 * i < BM) ? off : OOB);
        }
      } else {
        const unsigned abase = (unsigned)(((m0 + a_r) * (long long)g.Kd + k0 + a_kk) * 4);
        const bool kok = (MODE != MODE_GEMM) || (k0 + a_kk < g.Kd);
#pragma unroll
        for (int i = 0; i < MR16; ++i) {
          const unsigned off = abase + (unsigned)(i * 16 * g.Kd * 4);
          ra[i] = bload(rA, kok ? off : OOB);
        }
      }
    }
    if constexpr (MODE == MODE_GEMM) {
      const int k = k0 + a_kk;
#pragma unroll
      for (int i = 0; i < BROWS; ++i) {
        const long long n = n0 + a_r + 16 * i;
        rb[i] = bload(rB, (n < g.Ntot && k < g.Kd) ? (unsigned)((n * g.Kd + k) * 4) : OOB);
      }
    }
    if constexpr (MODE == MODE_CONV && KORD == KORD_TAP) {
      const int2 e = g.tab[k0 >> 4];  // one entry per chunk (scalar load)
      const int tap = e.y & 63;
      const unsigned half = tap < 32 ? mlo : mhi;
      const bool ok = (half >> (tap & 31)) & 1u;
      const unsigned voff = ok ? ((lbase + (unsigned)e.x) << 2) : OOB;
      okB = ok ? ~0u : 0u;
      const int cb = (e.y >> 8) + kg * BROWS;
#pragma unroll
      for (int q = 0; q < BROWS; ++q) {
        rb[q] = bload_s(rB, voff, (unsigned)q * g.sprod4);  // channels beyond C meet zero weights
        if constexpr (PRO == PRO_ACT) {
          const int ci = (cb + q < g.Cb) ? cb + q : g.Cb - 1;
          sp0[q] = g.pb[ci];
          sp1[q] = g.pb[g.Cb + ci];
        }
      }
    }
    if constexpr (MODE == MODE_CONV && KORD == KORD_CHAN) {
      okB = 0;
#pragma unroll
      for (int q = 0; q < BROWS; ++q) {
        const int2 e = g.tab[k0 + kg * BROWS + q];  // wave-uniform (scalar load); table is padded
        const int tap = e.y & 63;
        const unsigned half = tap < 32 ? mlo : mhi;
        const bool ok = (half >> (tap & 31)) & 1u;
        const unsigned off = ok ? ((lbase + (unsigned)e.x) << 2) : OOB;
        rb[q] = bload(rB, off);
        okB |= (ok ? 1u : 0u) << q;
        if constexpr (PRO == PRO_ACT) {  // wave-uniform scalar loads, consumed after the MFMAs of this chunk
          sp0[q] = g.pb[e.y >> 8];
          sp1[q] = g.pb[g.Cb + (e.y >> 8)];
        }
      }
    }
    if constexpr (MODE == MODE_WGRAD) {
      const unsigned p = wp + (unsigned)c * 16u;
      const bool pok = (long long)p < g.Ptot;
      const unsigned pp = pok ? p : 0u;
      const unsigned b = fdiv(pp, g.dPout);
      const unsigned rem = pp - b * (unsigned)Pout;
      const unsigned to = fdiv(rem, g.dHoWo);
      const unsigned r2 = rem - to * (unsigned)HoWo;
      const unsigned ho = fdiv(r2, g.dWo);
      const unsigned wo = r2 - ho * (unsigned)g.Wo;
      okA = pok ? 1u : 0u;
      okB = 0;
      // A[m][p] = dXout[b][m][p]   (rows >= M read garbage that is never stored)
      if constexpr (VA) {
        // 4 consecutive positions per lane (Pout % 4 == 0: a quad never straddles two samples)
        const unsigned pq = (wp - (unsigned)a_kk) + (unsigned)c * 16u + 4u * (unsigned)v_kq;
        const bool qok = (long long)pq < g.Ptot;
        const unsigned pqq = qok ? pq : 0u;
        const unsigned bq = fdiv(pqq, g.dPout);
        const unsigned remq = pqq - bq * (unsigned)Pout;
        const unsigned abase = (bq * (unsigned)g.Cout + (unsigned)(m0 + v_mr)) * (unsigned)Pout + remq;
#pragma unroll
        for (int i = 0; i < NP; ++i) {
          const unsigned off = (qok && v_mr + 64 * i < BM) ? ((abase + (unsigned)(64 * i) * (unsigned)Pout) << 2) : OOB;
          ra4[i] = bload4(rA, off);
        }
      } else {
        const unsigned abase = (b * (unsigned)g.Cout + (unsigned)(m0 + a_r)) * (unsigned)Pout + rem;
#pragma unroll
        for (int i = 0; i < MR16; ++i) {
          const unsigned off = pok ? ((abase + (unsigned)(16 * i) * (unsigned)Pout) << 2) : OOB;
          ra[i] = bload(rA, off);
        }
      }
      if constexpr (VB) {
        // quad of 4 consecutive output positions = 4 consecutive input elements of (b, ci, to*st + dt, :)
        const unsigned pq = (wp - (unsigned)a_kk) + (unsigned)c * 16u + 4u * (unsigned)v_kq;
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
        edgeLo = edgeHi = 0;
#pragma unroll
        for (int i = 0; i < NBQ; ++i) {
          const bool ok = qok && (unsigned)((int)toq * g.st + vb_dt[i]) < (unsigned)g.Ti &&
                          (VB == 1 || (unsigned)((int)hoq + vb_dh[i]) < (unsigned)g.Hi);
          const bool lo = VB == 2 && vb_dw[i] < 0 && woq == 0u;                       // element 0 is left padding
          const bool hi = VB == 2 && vb_dw[i] > 0 && woq + 4u == (unsigned)g.Wo;      // element 3 is right padding
          const unsigned el = xbq + (unsigned)vb_off[i] + (lo ? 1u : 0u) - (hi ? 1u : 0u);
          rb4[i] = bload4(rB, ok ? (el << 2) : OOB);
          okB |= (ok ? 1u : 0u) << i;
          edgeLo |= (lo ? 1u : 0u) << i;
          edgeHi |= (hi ? 1u : 0u) << i;
        }
        return;
      }
      // B[n][p] = act(X)[b][ci][in_pos(p, tap)]
      // table offsets / tap deltas already contain "- pad" (same table as the forward conv)
      const int ti0 = (int)to * g.st - 64, hi0 = (int)ho * g.sh - 64, wi0 = (int)wo * g.sw - 64;
      const unsigned xb = b * (unsigned)(g.Cin * THWi) + (unsigned)((ti0 + 64) * HWi + (hi0 + 64) * g.Wi + wi0 + 64);
#pragma unroll
      for (int i = 0; i < BROWS; ++i) {
        const int d = wt_d[i];
        const bool ok = pok && d >= 0 && (unsigned)(ti0 + (d & 255)) < (unsigned)g.Ti &&
                        (unsigned)(hi0 + ((d >> 8) & 255)) < (unsigned)g.Hi &&
                        (unsigned)(wi0 + ((d >> 16) & 255)) < (unsigned)g.Wi;
        rb[i] = bload(rB, ok ? ((xb + (unsigned)wt_off[i]) << 2) : OOB);
        okB |= (ok ? 1u : 0u) << i;
      }
    }
  };

  // ---------------------------------------------------------------- registers -> (mask, prologue) -> LDS
  auto store_chunk = [&](int buf, int c) __attribute__((always_inline)) {
    (void)c;
    float* As = smem + buf * (A_ELEMS + B_ELEMS);
    float* Bs = As + A_ELEMS;
    if constexpr (VA) {
#pragma unroll
      for (int i = 0; i < NP; ++i) {
        const int m = v_mr + 64 * i;
        if (m < BM) {
#pragma unroll
          for (int j = 0; j < 4; ++j) As[m * AS + 4 * v_kq + j] = ra4[i][j];
        }
      }
    }
    if constexpr (MODE == MODE_WGRAD) {
      if constexpr (!VA) {
#pragma unroll
        for (int i = 0; i < MR16; ++i) {
          const int m = a_r + 16 * i;
          As[m * AS + a_kk] = ra[i];
        }
      }
      if constexpr (VB) {
#pragma unroll
        for (int i = 0; i < NBQ; ++i) {
          const int nn = (tid >> 2) + 64 * i;
          const bool lo = (edgeLo >> i) & 1u, hi = (edgeHi >> i) & 1u;
          f32x4 q = rb4[i];
          if constexpr (PRO == PRO_ACT) {   // activation BEFORE the border shift: padding is zero after BN+ReLU
            const bool ok = (okB >> i) & 1u;
#pragma unroll
            for (int j = 0; j < 4; ++j) q[j] = ok ? apply_act(q[j], vb_s[i], vb_h[i], g.b_relu) : 0.f;
          }
          const f32x4 sh = lo ? (f32x4){0.f, q[0], q[1], q[2]} : (hi ? (f32x4){q[1], q[2], q[3], 0.f} : q);
#pragma unroll
          for (int j = 0; j < 4; ++j) Bs[nn * 18 + 4 * v_kq + j] = sh[j];
        }
      } else {
#pragma unroll
      for (int i = 0; i < BROWS; ++i) {
        const int nn = a_r + 16 * i;
        float v = rb[i];
        if constexpr (PRO == PRO_ACT) {
          v = apply_act(v, pBs[nn], pBs[BN + nn], g.b_relu);
          v = ((okB >> i) & 1u) ? v : 0.f;
        }
        Bs[nn * 18 + a_kk] = v;
      }
      }
    } else {
      if constexpr (!VA) {
#pragma unroll
        for (int i = 0; i < MR16; ++i) As[(a_r + 16 * i) * AS + a_kk] = ra[i];
      }
      if constexpr (MODE == MODE_GEMM) {
#pragma unroll
        for (int i = 0; i < BROWS; ++i) Bs[(a_r + 16 * i) * 18 + a_kk] = rb[i];
      } else {
        // per-k channel params are wave-uniform: short-lived scalar loads here (keeping them live
        // across the MFMA phase costs 16-40 registers)
#pragma unroll
        for (int q = 0; q < BROWS; ++q) {
          float v = rb[q];
          if constexpr (PRO == PRO_ACT) {
            v = apply_act(v, sp0[q], sp1[q], g.b_relu);  // params fetched with the data (no latency chain here)
            v = ((okB >> q) & 1u) ? v : 0.f;
          }
          Bs[(kg * BROWS + q) * BS + nl] = v;
        }
      }
    }
  };

  Acc acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < AR; ++r) acc[i][j][r] = 0.f;

  // fragment lane mapping: lane (fi, fk) holds A[row fi][k fk] and B[k fk][col fi] of an MFMA step
  const int fi = lane & (RT - 1), fk = lane / RT;
  const int mtv = __builtin_amdgcn_readfirstlane((mrem + RT - 1) / RT);  // valid row tiles (wave-uniform)
  // row of accumulator register r inside its tile
  auto rowof = [&](int r) __attribute__((always_inline)) { return MF ? (8 * (r >> 2) + 4 * fk + (r & 3)) : (fk * 4 + r); };

  auto compute = [&](int buf, auto full_tag) __attribute__((always_inline)) {
    constexpr bool FULL = decltype(full_tag)::value;
    const float* As = smem + buf * (A_ELEMS + B_ELEMS);
    const float* Bs = As + A_ELEMS;
#pragma unroll
    for (int kk = 0; kk < 16 / KPS; ++kk) {
      float a[MT], b[NT];
#pragma unroll
      for (int i = 0; i < MT; ++i) a[i] = As[(i * RT + fi) * AS + kk * KPS + fk];
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        if constexpr (BKF) b[j] = Bs[((wave * NT + j) * RT + fi) * 18 + kk * KPS + fk];
        else b[j] = Bs[(kk * KPS + fk) * BS + (wave * NT + j) * RT + fi];
      }
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        if (FULL || i < mtv) {  // wave-uniform: skip row tiles beyond M in ragged blocks
#pragma unroll
          for (int j = 0; j < NT; ++j)
            if constexpr (MF) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
            else acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
      }
    }
  };

  // ---------------------------------------------------------------- main loop
  auto main_loop = [&](auto full_tag) __attribute__((always_inline)) {
    load_chunk(0);
    store_chunk(0, 0);
    __syncthreads();
    // steady state is ONE basic block (no conditionals): the compiler interleaves the next chunk's
    // loads, this chunk's MFMAs and the LDS writes; any branch in here costs 30-40 % (measured).
    for (int c = 0; c + 1 < nchunks; ++c) {
      load_chunk(c + 1);
      compute(c & 1, full_tag);
      store_chunk((c + 1) & 1, c + 1);
      __syncthreads();
    }
    compute((nchunks - 1) & 1, full_tag);
    __syncthreads();
  };
  if (nchunks > 0) {
    if (mtv >= MT) main_loop(std::true_type{});
    else main_loop(std::false_type{});
  }

  // ---------------------------------------------------------------- epilogue
  // accumulator layout: row = i*RT + rowof(r), col = (wave*NT + j)*RT + fi
  if constexpr (MODE == MODE_CONV) {
    const int dP = g.D0 * g.D1 * g.D2;
    float* Cp = g.C + (size_t)split * (size_t)g.split_stride;  // split-K: slice-private partial output
    size_t obase[NT];
    bool cok[NT];
    const int NQ = g.Q0 * g.Q1 * g.Q2, Q12 = g.Q1 * g.Q2;
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const long long n = n0 + (wave * NT + j) * RT + fi;
      cok[j] = n < g.Ntot;
      const long long nn = cok[j] ? n : 0;
      const int b = (int)(nn / NQ);
      int rem = (int)(nn - (long long)b * NQ);
      const int q0 = rem / Q12;
      rem -= q0 * Q12;
      const int q1 = rem / g.Q2, q2 = rem - q1 * g.Q2;
      obase[j] = (size_t)b * g.M * dP + (size_t)(q0 * g.dmul0 + g.dorg0) * (g.D1 * g.D2) +
                 (size_t)(q1 * g.dmul1 + g.dorg1) * g.D2 + (size_t)(q2 * g.dmul2 + g.dorg2);
    }
    float* red = smem;            // [2][4 waves][BM]   (the main loop ended with a barrier)
    if constexpr (EPI == EPI_PLAIN) {
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        if (i < mtv) {
#pragma unroll
          for (int r = 0; r < AR; ++r) {
            const int m = i * RT + rowof(r);
            if (m < mrem) {
#pragma unroll
              for (int j = 0; j < NT; ++j) {
                if (cok[j]) {
                  const size_t ad = obase[j] + (size_t)(m0 + m) * dP;
                  float v = acc[i][j][r];
                  if (g.E) v += g.E[ad];
                  Cp[ad] = v;
                }
              }
            }
          }
        }
      }
    } else {
    float* rpar = smem + 8 * BM;  // [4][BM] s, h, mean, invstd of this block's rows
    constexpr bool bnr = true;
    if (bnr) {
      for (int m = tid; m < BM; m += 256) {
        const int mm = (m0 + m < g.M) ? m0 + m : g.M - 1;
        rpar[m] = g.rss[mm];
        rpar[BM + m] = g.rss[g.M + mm];
        rpar[2 * BM + m] = g.rmi[mm];
        rpar[3 * BM + m] = g.rmi[g.M + mm];
      }
      __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      if (i < mtv) {
        // all loads of this 16-row tile first (they are independent; the stores below would otherwise
        // fence them one by one), then the arithmetic and the stores
        float xv[AR][NT], ev[AR][NT];
#pragma unroll
        for (int r = 0; r < AR; ++r) {
          const int m = i * RT + rowof(r);
#pragma unroll
          for (int j = 0; j < NT; ++j) {
            const bool ok = (m < mrem) && cok[j];
            const size_t ad = obase[j] + (size_t)(m0 + m) * dP;
            xv[r][j] = (bnr && ok) ? g.R[ad] : 0.f;
            ev[r][j] = (g.E && ok) ? g.E[ad] : 0.f;
          }
        }
#pragma unroll
        for (int r = 0; r < AR; ++r) {
          const int m = i * RT + rowof(r);
          const bool mok = m < mrem;
          float ps = 0.f, ph = 0.f, pm = 0.f, pi = 0.f, s0 = 0.f, s1 = 0.f;
          if (bnr) { ps = rpar[m]; ph = rpar[BM + m]; pm = rpar[2 * BM + m]; pi = rpar[3 * BM + m]; }
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
          if (bnr) {  // wave-uniform; all lanes take part in the shuffles
#pragma unroll
            for (int o = 1; o < RT; o <<= 1) {
              s0 += __shfl_xor(s0, o, 64);
              s1 += __shfl_xor(s1, o, 64);
            }
            if (fi == 0) {
              red[wave * BM + m] = s0;
              red[(4 + wave) * BM + m] = s1;
            }
          }
        }
      }
    }
    if (bnr) {
      __syncthreads();
      for (int m = tid; m < BM; m += 256) {
        if (m < mrem) {
          const float a = ((red[m] + red[BM + m]) + red[2 * BM + m]) + red[3 * BM + m];
          const float b = ((red[4 * BM + m] + red[5 * BM + m]) + red[6 * BM + m]) + red[7 * BM + m];
          float* o = g.rpart + ((size_t)(m0 + m) * g.rslots + g.rslot0 + nblk) * 2;
          o[0] = a;
          o[1] = b;
        }
      }
    }
    }
    if (g.stat_sum) {
      // per-channel partial statistics of this block's columns (fixed order -> deterministic)
#pragma unroll
      for (int i = 0; i < MT; ++i) {
#pragma unroll
        for (int r = 0; r < AR; ++r) {
          float s = 0.f, q = 0.f;
#pragma unroll
          for (int j = 0; j < NT; ++j) {
            const float v = cok[j] ? acc[i][j][r] : 0.f;
            s += v;
            q += v * v;
          }
#pragma unroll
          for (int o = 1; o < RT; o <<= 1) {
            s += __shfl_xor(s, o, 64);
            q += __shfl_xor(q, o, 64);
          }
          if (fi == 0) {
            const int m = i * RT + rowof(r);
            red[wave * BM + m] = s;
            red[(4 + wave) * BM + m] = q;
          }
        }
      }
      __syncthreads();
      for (int m = tid; m < BM; m += 256) {
        if (m < mrem) {
          const float s = ((red[m] + red[BM + m]) + red[2 * BM + m]) + red[3 * BM + m];
          const float q = ((red[4 * BM + m] + red[5 * BM + m]) + red[6 * BM + m]) + red[7 * BM + m];
          g.stat_sum[(size_t)(m0 + m) * g.nblkN + nblk] = s;
          g.stat_sq[(size_t)(m0 + m) * g.nblkN + nblk] = q;
        }
      }
    }
  } else {
    // WGRAD / GEMM: C[(split)][m][n], row-major with leading dimension ldc
    float* Cout_ = g.C + (MODE == MODE_WGRAD ? (size_t)split * g.M * g.ldc : 0);
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      if (i < mtv) {
#pragma unroll
        for (int r = 0; r < AR; ++r) {
          const int m = i * RT + rowof(r);
          if (m < mrem) {
#pragma unroll
            for (int j = 0; j < NT; ++j) {
              const long long n = n0 + (wave * NT + j) * RT + fi;
              if (n < g.Ntot) {
                float v = acc[i][j][r];
                if constexpr (MODE == MODE_GEMM) {
                  if (g.bias) v += g.bias[n];
                }
                Cout_[(size_t)(m0 + m) * g.ldc + n] = v;
              }
            }
          }
        }
      }
    }
  }
}

// SUBSET: which MODE_CONV variants a translation unit instantiates (the forward conv never uses the BNR epilogue,
// the backward-data conv never has an operand prologue) -- keeps the per-file compile time down.
enum { SUB_ALL = 0, SUB_FWD = 1, SUB_DGRAD = 2 };

template <int MODE, int MT, int NT, int MF = 0, int SUBSET = SUB_ALL>
inline void launch_igemm(const IgemmArgs& a, int splits, bool vec_a, hipStream_t st) {
  dim3 grid(a.nblkM * a.nblkN * splits, 1, 1);
  const bool act = (MODE != MODE_GEMM) && a.b_pro == PRO_ACT;
  if constexpr (MODE == MODE_WGRAD) {
    if (a.vec_b == 1 && vec_a) {   // 16-byte loads for both operands: temporal / pointwise convs
      if (act) hipLaunchKernelGGL((igemm_kernel<MODE, MT, NT, true, PRO_ACT, KORD_CHAN, EPI_PLAIN, MF, 1>), grid, dim3(256), 0, st, a);
      else hipLaunchKernelGGL((igemm_kernel<MODE, MT, NT, true, PRO_NONE, KORD_CHAN, EPI_PLAIN, MF, 1>), grid, dim3(256), 0, st, a);
      return;
    }
    if (a.vec_b == 2 && vec_a) {   // ... same-padded 3x3 spatial taps
      if (act) hipLaunchKernelGGL((igemm_kernel<MODE, MT, NT, true, PRO_ACT, KORD_CHAN, EPI_PLAIN, MF, 2>), grid, dim3(256), 0, st, a);
      else hipLaunchKernelGGL((igemm_kernel<MODE, MT, NT, true, PRO_NONE, KORD_CHAN, EPI_PLAIN, MF, 2>), grid, dim3(256), 0, st, a);
      return;
    }
  }
#define SLV_K(VA_, PRO_, KORD_, EPI_) \
  hipLaunchKernelGGL((igemm_kernel<MODE, MT, NT, VA_, PRO_, KORD_, EPI_, MF>), grid, dim3(256), 0, st, a)
  if constexpr (MODE == MODE_CONV) {
    if constexpr (SUBSET != SUB_FWD) {
      if (a.R) {  // backward-data with the fused BatchNorm-backward reduction (never has an operand prologue)
        if (a.kord == KORD_TAP) SLV_K(true, PRO_NONE, KORD_TAP, EPI_BNR);
        else if (vec_a) SLV_K(true, PRO_NONE, KORD_CHAN, EPI_BNR);
        else SLV_K(false, PRO_NONE, KORD_CHAN, EPI_BNR);
        return;
      }
    }
    // tap-major K: Kd is a multiple of 16 and A is 64-byte aligned -> always vector A loads
    if constexpr (SUBSET != SUB_DGRAD) {
      if (act) {
        if (a.kord == KORD_TAP) SLV_K(true, PRO_ACT, KORD_TAP, EPI_PLAIN);
        else if (vec_a) SLV_K(true, PRO_ACT, KORD_CHAN, EPI_PLAIN);
        else SLV_K(false, PRO_ACT, KORD_CHAN, EPI_PLAIN);
        return;
      }
    }
    if (a.kord == KORD_TAP) SLV_K(true, PRO_NONE, KORD_TAP, EPI_PLAIN);
    else if (vec_a) SLV_K(true, PRO_NONE, KORD_CHAN, EPI_PLAIN);
    else SLV_K(false, PRO_NONE, KORD_CHAN, EPI_PLAIN);
  } else if constexpr (MODE == MODE_GEMM) {
    if (vec_a) SLV_K(true, PRO_NONE, KORD_CHAN, EPI_PLAIN); else SLV_K(false, PRO_NONE, KORD_CHAN, EPI_PLAIN);
  } else {   // MODE_WGRAD without the 16-byte gathered-operand path
    if (vec_a) { if (act) SLV_K(true, PRO_ACT, KORD_CHAN, EPI_PLAIN); else SLV_K(true, PRO_NONE, KORD_CHAN, EPI_PLAIN); }
    else { if (act) SLV_K(false, PRO_ACT, KORD_CHAN, EPI_PLAIN); else SLV_K(false, PRO_NONE, KORD_CHAN, EPI_PLAIN); }
  }
#undef SLV_K
}

// choose the row-tile: returns MT for a given M (rows) -- see header comment
inline int pick_mt(int M) {
  if (M <= 64) return 4;
  if (M <= 128) return 8;
  if (M <= 144) return 9;
  if (M <= 240) return 15;
  // multi-block: minimise padded rows, prefer the larger tile on ties
  const int cand[4] = {15, 9, 8, 4};
  int best = 8;
  long bestw = 1L << 60;
  for (int c : cand) {
    const int bm = c * 16;
    const long padded = (long)((M + bm - 1) / bm) * bm;
    if (padded < bestw) { bestw = padded; best = c; }
  }
  return best;
}

}  // namespace slv
