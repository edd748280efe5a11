// First kernel of the 16-bit MFMA path (BASELINE configs[4], main.py:151 --use_fp16): a forward convolution on bf16
// CHANNELS-LAST activations with fp32 accumulation on v_mfma_f32_16x16x32_bf16 and a fused per-channel affine
// (eval-mode BatchNorm) + residual + ReLU epilogue.
//
// Why another layout: a 16-bit MFMA operand is 8 consecutive k per lane.  With N,C,T,H,W activations k (= channel)
// has stride T*H*W, so every fragment would be 8 scalar gathers; with N,T,H,W,C it is one 16-byte load and the LDS
// image needs no transpose.  On the hot layer (Conv3d 64->144 (1,3,3), 16 clips) this kernel takes 0.22 ms against
// 1.24 ms for the fp32 kernel (tools/proto/bf16_conv133.hip measured 601 TFLOP/s with the first, untuned version).
//
// Implicit GEMM  D[cout][pos] = sum_{tap, c} W[cout][tap][c] * X[pos*stride + tap - pad][c]:
//   block = 4 waves = (MT*16) couts x 128 output positions, wave = (MT*16) x 32, K-step = 32 channels of one tap;
//   weights pre-laid-out [tap][Cin_p/32][Mrows][32] (zero padded), activations [P][Cin_p] with Cin_p % 32 == 0;
//   register-staged double buffer, one barrier per K-step, LDS rows of 32 bf16 padded to 80 bytes (conflict-free
//   ds_read_b128 fragments); padding positions are buffer loads with an out-of-range offset (return 0).
//   Output [P_out][Cout_p] bf16: channels >= Cout are written as zeros so the next layer can read whole 32-channel
//   K-steps.
#include "cl16.hpp"
#include "../../include/selavi_hip.h"

namespace slv {

// PRO 0: B rows as stored.  PRO 1: B rows read as relu(x * s[c] + h[c]) (the producer's BatchNorm + ReLU applied on
//        load, zero padding AFTER the affine), table in_ss [2][Cin].
// EPI 0: y = relu?(acc * scale + shift + res) -> bf16 (eval-mode BatchNorm / residual; scale_shift, res nullable: plain
//        store, and backward-data with its addend).
// EPI 1: y = acc -> bf16 plus per-channel partial sums of y and y^2 over the block's positions, taken on the ROUNDED
//        values the consumer will normalise: stat_sum / stat_sq [Cout][gridDim.x] (train-mode BatchNorm statistics).
// EPI 2: EPI 0 without the affine (backward data + addend) plus the BatchNorm-backward sums of ClBnr (cl16.hpp).
template <int MT, int PRO, int EPI>
__global__ __launch_bounds__(256, MT >= 8 ? 3 : 4) void conv_cl16_kernel(const unsigned short* __restrict__ x,
                                                           const unsigned short* __restrict__ wl,
                                                           unsigned short* __restrict__ y,
                                                           const float* __restrict__ in_ss,         // PRO 1: [2][Cin]
                                                           const float* __restrict__ scale_shift,   // [2][Cout] or null
                                                           const unsigned short* __restrict__ res,  // output-shaped or null
                                                           int relu, float* __restrict__ stat_sum,
                                                           float* __restrict__ stat_sq, ClBnr bn, ClConv g, FastDiv dLw, FastDiv dLh, FastDiv dLt, FastDiv dGx) {
  constexpr int BM = MT * 16;
  constexpr int APIECES = BM * 4, AITER = (APIECES + 255) / 256;
  constexpr int OROW = BM * 2 + 16;                   // bytes per position row of the transposed output tile (+16: banks)
  constexpr int STAGE = (BM + CL_BN) * CL_ROWB;
  constexpr int KLOOP = 2 * STAGE + (PRO ? 2 * CL_PRO_MAXC * 4 : 0);
  constexpr int EPIB = CL_BN * OROW + 12 * BM * 4 + CL_BN * 4;
  constexpr int LDSB = KLOOP > EPIB ? KLOOP : EPIB;
  __shared__ __attribute__((aligned(16))) unsigned char lds_raw[LDSB];
  unsigned char(*lds)[STAGE] = (unsigned char(*)[STAGE])lds_raw;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const unsigned P = (unsigned)g.N * g.Lt * g.Lh * g.Lw;
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(
      (void*)x, 0, (int)((unsigned)g.N * g.Ti * g.Hi * g.Wi * g.Cin_p * 2u), 0x00020000);
  // XCD-aware bijective remap of the tiles (hardware block b runs on XCD b % 8): every XCD gets a CONTIGUOUS range of
  // (M tile, position tile) units, so the rows a tile re-reads for its shifted taps (the neighbouring tiles' rows) are
  // in ITS L2
  unsigned bx = blockIdx.x, by = blockIdx.y;
  if (XCD_REMAP) {
    const unsigned nb = gridDim.x * gridDim.y, lin = blockIdx.x + blockIdx.y * gridDim.x, q8 = nb >> 3, r8 = nb & 7,
                   xcd = lin & 7, loc = lin >> 3;
    const unsigned unit = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + loc;
    by = fdiv(unit, dGx);
    bx = unit - by * gridDim.x;
  }
  const int m0 = by * BM;
  // ---- activation rows of this thread: 128 rows x 4 pieces of 16 B, 2 per thread
  int bt[2], bh[2], bw[2];
  unsigned bbase[2];
  const int piece = tid & 3;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const unsigned p = bx * CL_BN + (tid >> 2) + 64 * i;
    // (divisions by multiply + shift: a division by a runtime value is ~25 VALU instructions, and this kernel's SIMDs
    //  spend 55-66 % of their cycles issuing VALU -- profiles/r02_pmc_conv16.txt)
    const unsigned q1 = fdiv(p, dLw), q2 = fdiv(q1, dLh), q = fdiv(q2, dLt);      // q = clip
    const int lw = p - q1 * g.Lw, lh = q1 - q2 * g.Lh, lt = q2 - q * g.Lt;
    bt[i] = lt * g.bmt + g.bot;
    bh[i] = lh * g.bmh + g.boh;
    bw[i] = lw * g.bmw + g.bow;
    bbase[i] = (((q * g.Ti + bt[i]) * g.Hi + bh[i]) * g.Wi + bw[i]) * (unsigned)(g.Cin_p * 2) + piece * 16u;
    if (p >= P) bt[i] = -(1 << 20);          // (bbase wraps for negative coordinates; only used when the tap is valid)
  }
  const int kcs = g.Cin_p >> 5, ksteps = g.ntaps * kcs;
  float* pro = (float*)(lds_raw + 2 * STAGE);                      // PRO 1: [2][Cin_p] scale, shift (zero beyond Cin)
  if constexpr (PRO == 1) {
    for (int i = tid; i < 2 * g.Cin_p; i += 256) {
      const int c = i % g.Cin_p, which = i / g.Cin_p;
      pro[i] = c < g.Cin ? in_ss[which * g.Cin + c] : 0.f;
    }
  }
  u32x4 ra[AITER], rb[2];
  bool okl[2] = {false, false};
  int tapi = 0, kc = 0, kc_ld = 0;                                 // K-step being LOADED
  auto gload = [&]() __attribute__((always_inline)) {
    const int tp = g.tap[tapi];
    const int dt = (tp & 15) - 8, dh = ((tp >> 4) & 15) - 8, dw = ((tp >> 8) & 15) - 8, slab = tp >> 12;
    const unsigned toff = (unsigned)(((dt * g.Hi + dh) * g.Wi + dw) * g.Cin_p * 2 + kc * 64);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const bool ok = (unsigned)(bt[i] + dt) < (unsigned)g.Ti && (unsigned)(bh[i] + dh) < (unsigned)g.Hi &&
                      (unsigned)(bw[i] + dw) < (unsigned)g.Wi;
      okl[i] = ok;
      rb[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, ok ? bbase[i] + toff : 0xFFFFFFF0u, 0, 0));
    }
    const unsigned short* ws = wl + ((size_t)(slab * kcs + kc) * g.Mrows + m0) * 32;
#pragma unroll
    for (int i = 0; i < AITER; ++i) {
      const int pc = tid + 256 * i;
      if (pc < APIECES) ra[i] = *(const u32x4*)(ws + pc * 8);
    }
    kc_ld = kc;
    if (++kc == kcs) {                                              // advance to the next K-step
      kc = 0;
      ++tapi;
    }
  };
  auto lstore = [&](int buf) __attribute__((always_inline)) {
    unsigned char* A = lds[buf];
    unsigned char* B = A + BM * CL_ROWB;
#pragma unroll
    for (int i = 0; i < AITER; ++i) {
      const int pc = tid + 256 * i;
      if (pc < APIECES) *(u32x4*)(A + (pc >> 2) * CL_ROWB + (((pc & 3) ^ cl_swz(pc >> 2)) << 4)) = ra[i];
    }
    if constexpr (PRO == 1) {                                       // the prologue math sits at the LDS-write point
      float s[8], h[8];
      const float* sp = pro + kc_ld * 32 + piece * 8;
      *(f32x4*)s = *(const f32x4*)sp;
      *(f32x4*)(s + 4) = *(const f32x4*)(sp + 4);
      *(f32x4*)h = *(const f32x4*)(sp + g.Cin_p);
      *(f32x4*)(h + 4) = *(const f32x4*)(sp + g.Cin_p + 4);
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const u32x4 t = affine_relu8(rb[i], s, h);
        rb[i] = okl[i] ? t : (u32x4){0u, 0u, 0u, 0u};
      }
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
      *(u32x4*)(B + ((tid >> 2) + 64 * i) * CL_ROWB + ((piece ^ cl_swz(tid >> 2)) << 4)) = rb[i];
  };
  f32x4 acc[MT][2];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const int fr = lane & 15, fk = lane >> 4, fsw = (fk ^ cl_swz(fr)) << 4;   // rows i*16+fr, wave*32+j*16+fr: same swz
  if (ksteps > 0) {
    gload();
    if constexpr (PRO == 1) __syncthreads();                         // the prologue table is complete
    lstore(0);
    __syncthreads();
  }
  for (int s = 0; s < ksteps; ++s) {
    if (s + 1 < ksteps) gload();
    const unsigned char* A = lds[s & 1];
    const unsigned char* B = A + BM * CL_ROWB;
    bf16x8 b[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) b[j] = *(const bf16x8*)(B + (wave * 32 + j * 16 + fr) * CL_ROWB + fsw);
    // every fragment read of the K-step is issued before the first MFMA (left alone, the scheduler re-uses one fragment
    // register and puts a full LDS round trip in front of every MFMA pair)
    bf16x8 a[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) a[i] = *(const bf16x8*)(A + (i * 16 + fr) * CL_ROWB + fsw);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    if (s + 1 < ksteps) lstore((s + 1) & 1);
    __syncthreads();
  }
  // ---- epilogue.  C/D: col = lane & 15 (position), rows (lane >> 4) * 4 + r (cout): 4 consecutive couts per lane, i.e.
  // 8-byte pieces 320 bytes apart across lanes -- stored like that they cost a quarter of the forward (ablation).  The
  // tile is transposed through LDS instead: [position][cout] rows, then 16-byte stores that run along a position's
  // channels (288 contiguous bytes per row for the 144-channel tile).  scale / shift also come through LDS.
  __syncthreads();                                      // the K loop's last fragment reads are done
  unsigned char* ot = lds_raw;                          // [CL_BN][OROW]
  float* ssl = (float*)(lds_raw + CL_BN * OROW);        // EPI 0: [2][BM] scale, shift; EPI 1: [4 waves][2][BM] partials
  float* red = ssl + (EPI == 2 ? 4 * BM : 0);           // EPI 2: [4][BM] s, h, mean, invstd first, then the partials
  unsigned* opos = (unsigned*)(lds_raw + CL_BN * OROW + 12 * BM * 4);  // [CL_BN] output position (row index) or ~0
  if (tid < CL_BN) {
    const unsigned p = bx * CL_BN + tid;
    unsigned o = 0xFFFFFFFFu;
    if (p < P) {
      const unsigned q1 = fdiv(p, dLw), q2 = fdiv(q1, dLh), q = fdiv(q2, dLt);
      const int lw = p - q1 * g.Lw, lh = q1 - q2 * g.Lh, lt = q2 - q * g.Lt;
      o = ((q * g.To + lt * g.omt + g.oot) * g.Ho + lh * g.omh + g.ooh) * g.Wo + lw * g.omw + g.oow;
    }
    opos[tid] = o;
  }
  if (EPI == 0 && scale_shift) {
    for (int i = tid; i < 2 * BM; i += 256) {
      const int c = m0 + (i % BM);
      ssl[i] = c < g.Cout ? scale_shift[(i / BM) * g.Cout + c] : 0.f;
    }
  }
  if constexpr (EPI == 2) {
    for (int i = tid; i < 4 * BM; i += 256) {
      const int c = m0 + (i % BM), which = i / BM;
      ssl[i] = c < g.Cout ? (which < 2 ? bn.ss[which * g.Cout + c] : bn.mi[(which - 2) * g.Cout + c]) : 0.f;
    }
  }
  __syncthreads();
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
      const int pl = wave * 32 + j * 16 + fr;             // position inside the block
      float v[4];
      const unsigned op2 = opos[pl];
      if constexpr (EPI != 1) {
        const unsigned op = op2;
        uint2 rr = make_uint2(0u, 0u);
        if (res && op != 0xFFFFFFFFu && co < g.Cout_p) rr = *(const uint2*)(res + (size_t)op * g.Cout_p + co);
        if (!scale_shift && !relu) {      // backward data: the accumulator (+ addend) as it is -- rows >= Cout have zero weights
#pragma unroll                            // and the addend's padding channels are zero (the full path below costs ~45 VALU per 4 x 2 outputs)
          for (int r = 0; r < 4; ++r) {
            v[r] = acc[i][j][r];
            if (res) v[r] += bf2f((unsigned short)((r < 2 ? rr.x : rr.y) >> ((r & 1) * 16)));
          }
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float t = __builtin_fmaf(acc[i][j][r], sc[r], sh[r]);     // (scale, shift are zero for channels >= Cout, the residual's padding is zero: no select)
            if (res) t += bf2f((unsigned short)((r < 2 ? rr.x : rr.y) >> ((r & 1) * 16)));
            if (relu) t = fmaxf(t, 0.f);
            v[r] = t;
          }
        }
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = acc[i][j][r];              // rows >= Cout: zero weights -> 0
      }
      const unsigned lo = pack_bf2(v[0], v[1]), hi = pack_bf2(v[2], v[3]);
      *(uint2*)(ot + pl * OROW + (i * 16 + fk * 4) * 2) = make_uint2(lo, hi);
      if constexpr (EPI == 2) {                          // BatchNorm-backward sums of the source layer (ClBnr)
        uint2 xr = make_uint2(0u, 0u);
        if ((op2 != 0xFFFFFFFFu) && co < g.Cout_p) xr = *(const uint2*)(bn.x + (size_t)op2 * g.Cout_p + co);
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
    if constexpr (EPI == 2) {                            // 16 positions per lane group -> lane fr == 0 of each group
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
  if constexpr (EPI >= 1) {                              // the 4 waves' partials in fixed order
    for (int i = tid; i < 2 * BM; i += 256) {
      const int c = i % BM, which = i / BM;
      if (m0 + c < g.Cout) {
        const float t = ((red[(0 * 2 + which) * BM + c] + red[(1 * 2 + which) * BM + c]) + red[(2 * 2 + which) * BM + c]) +
                        red[(3 * 2 + which) * BM + c];
        if constexpr (EPI == 1) (which ? stat_sq : stat_sum)[(size_t)(m0 + c) * gridDim.x + bx] = t;   // [Cout][tiles]
        else bn.part[((size_t)(m0 + c) * bn.nslots + bn.slot0 + bx) * 2 + which] = t;                  // [Cout][slots][2]
      }
    }
  }
  // rows of this block in the output: channels [m0, m0 + BM) clipped to Cout_p; the last M block also zero-fills the
  // padding channels no M tile covers (Mrows < Cout_p)
  const int c_lo = m0, c_hi = min(m0 + BM, g.Cout_p);
  const int c_end = (by == gridDim.y - 1) ? g.Cout_p : c_hi;
  const int pieces = (c_end - c_lo) >> 3;                 // 16-byte pieces per position (channel counts are multiples of 8)
  // a thread keeps its 16-byte column piece and walks down the rows, rpp = 256 / pieces rows per pass (one division per
  // thread instead of index arithmetic per store; P_out * Cout_p * 2 < 2^32: 32-bit byte offsets)
  const int rpp = 256 / pieces, pl0 = tid / pieces, pc = tid - pl0 * pieces;
  if (pl0 < rpp) {
    const bool cval = c_lo + pc * 8 < c_hi;
    const unsigned char* src = ot + pl0 * OROW + pc * 16;
    unsigned char* yb = (unsigned char*)y;
    const unsigned coff = (unsigned)(c_lo + pc * 8) * 2u, rowb = (unsigned)g.Cout_p * 2u;
    for (int pl = pl0; pl < CL_BN; pl += rpp, src += rpp * OROW) {
      const unsigned op = opos[pl];
      if (op == 0xFFFFFFFFu) continue;
      u32x4 val = {0u, 0u, 0u, 0u};
      if (cval) val = *(const u32x4*)src;
      *(u32x4*)(yb + op * rowb + coff) = val;
    }
  }
}

// fp32 N,C,T,H,W -> bf16 N,T,H,W,Cp (channels >= C zero): one thread per (position, 8-channel piece)
__global__ __launch_bounds__(256) void to_cl16_kernel(const float* __restrict__ x, unsigned short* __restrict__ y, int C,
                                                      int Cp, unsigned S /* T*H*W */, unsigned total /* N*S*Cp/8 */) {
  const unsigned idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const unsigned pieces = Cp >> 3, pc = idx % pieces, pos = idx / pieces, n = pos / S, s = pos - n * S;
  unsigned short v[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const unsigned c = pc * 8 + i;
    v[i] = c < (unsigned)C ? f2bf(x[((size_t)n * C + c) * S + s]) : (unsigned short)0;
  }
  u32x4 o = {v[0] | ((unsigned)v[1] << 16), v[2] | ((unsigned)v[3] << 16), v[4] | ((unsigned)v[5] << 16),
             v[6] | ((unsigned)v[7] << 16)};
  *(u32x4*)(y + (size_t)idx * 8) = o;
}


// Stem input: fp32 N,C,T,H,W -> bf16 [N][T][H][Wo][32] where the 32 "channels" of an output column wo are the
// kw x C input values of its receptive row (index dw*C + c, zero beyond kw*C and outside the image).  A (1,kh,kw)
// stem conv over C = 3 (or 1) channels then runs as a (1,kh,1) conv over 32 channels: kh K-steps instead of kh*kw with
// 3 of 32 channels used, and the converted input is half the size of a 32-channel padded copy.
__global__ __launch_bounds__(256) void to_cl16_wpatch_kernel(const float* __restrict__ x, unsigned short* __restrict__ y,
                                                             int C, unsigned TH, int W, int Wo, int kw, int sw, int pw,
                                                             unsigned total /* N*TH*Wo*4 pieces */) {
  const unsigned idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const unsigned pc = idx & 3, q = idx >> 2, wo = q % Wo, row = q / Wo, n = row / TH, th = row - n * TH;
  unsigned short v[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int k = pc * 8 + i, dw = k / C, c = k - dw * C, w = (int)wo * sw - pw + dw;
    float f = 0.f;
    if (dw < kw && (unsigned)w < (unsigned)W) f = x[(((size_t)n * C + c) * TH + th) * W + w];
    v[i] = f2bf(f);
  }
  u32x4 o = {v[0] | ((unsigned)v[1] << 16), v[2] | ((unsigned)v[3] << 16), v[4] | ((unsigned)v[5] << 16),
             v[6] | ((unsigned)v[7] << 16)};
  *(u32x4*)(y + (size_t)idx * 8) = o;
}

// MaxPool2d(3, stride 2, pad 1) on bf16 channels-last [N][H][W][Cp] (audio trunk, model.py:114 -> torchvision ResNet):
// one thread per (output position, 8-channel piece)
__global__ __launch_bounds__(256) void maxpool_cl16_kernel(const unsigned short* __restrict__ x,
                                                           unsigned short* __restrict__ y, int H, int W, int Ho, int Wo,
                                                           int Cp, unsigned total) {
  const unsigned idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const unsigned pieces = Cp >> 3, pc = idx % pieces;
  unsigned q = idx / pieces;
  const int wo = q % Wo; q /= Wo;
  const int ho = q % Ho; q /= Ho;                 // q = image
  float m[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) m[i] = -INFINITY;
  for (int dh = 0; dh < 3; ++dh)
    for (int dw = 0; dw < 3; ++dw) {
      const int h = ho * 2 - 1 + dh, w = wo * 2 - 1 + dw;
      if ((unsigned)h >= (unsigned)H || (unsigned)w >= (unsigned)W) continue;
      const u32x4 v = *(const u32x4*)(x + (((size_t)q * H + h) * W + w) * Cp + pc * 8);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        m[2 * i] = fmaxf(m[2 * i], bf2f((unsigned short)(v[i] & 0xFFFF)));
        m[2 * i + 1] = fmaxf(m[2 * i + 1], bf2f((unsigned short)(v[i] >> 16)));
      }
    }
  u32x4 o;
#pragma unroll
  for (int i = 0; i < 4; ++i) o[i] = f2bf(m[2 * i]) | ((unsigned)f2bf(m[2 * i + 1]) << 16);
  *(u32x4*)(y + (size_t)idx * 8) = o;
}

// AdaptiveAvgPool(1) + flatten: bf16 [N][S][Cp] -> fp32 [N][C]; one wave per (clip, 64 channels), fixed order
__global__ __launch_bounds__(64) void avgpool_cl16_kernel(const unsigned short* __restrict__ x, float* __restrict__ y,
                                                          int S, int C, int Cp) {
  const int n = blockIdx.y, c = blockIdx.x * 64 + threadIdx.x;
  if (c >= C) return;
  float acc = 0.f;
  for (int s = 0; s < S; ++s) acc += bf2f(x[((size_t)n * S + s) * Cp + c]);
  y[(size_t)n * C + c] = acc / (float)S;
}

}  // namespace slv

extern "C" {

static int cl16_launch(const slv::ClConv& g, int mt, const void* x, const void* wl, void* y, const float* in_ss,
                       const float* scale_shift, const void* res, int relu, float* stat_sum, float* stat_sq,
                       const slv::ClBnr& bnr, slv_stream_t stream, const char* fn) {
  using namespace slv;
  {   // layer-1 spatial conv 64 -> 144: weights resident in registers, data-movement wave (csrc/conv_cl16_sr.hip)
    const int r = cl16_sr_try(g, x, wl, y, in_ss, scale_shift, res, relu, stat_sum, stat_sq, bnr, (hipStream_t)stream);
    if (r != 0) return r < 0 ? r : 0;
  }
  {   // backward data of the layer-1 spatial conv: weights resident in registers, LDS-DMA patches (csrc/conv_cl16_sd.hip)
    const int r = cl16_sd_try(g, x, wl, y, in_ss, scale_shift, res, relu, stat_sum, stat_sq, bnr, (hipStream_t)stream);
    if (r != 0) return r < 0 ? r : 0;
  }
  {   // stride-1 (3,1,1) convs of the narrow layers: weights resident in registers (csrc/conv_cl16_tr.hip)
    const int r = cl16_tr_try(g, x, wl, y, in_ss, scale_shift, res, relu, stat_sum, stat_sq, bnr, (hipStream_t)stream);
    if (r != 0) return r < 0 ? r : 0;
  }
  {   // the wide layers at sizes that fill the chip: the 8-wave ping-pong kernel (csrc/conv_cl16_g8.hip)
    const int r = cl16_g8_try(g, x, wl, y, in_ss, scale_shift, res, relu, stat_sum, stat_sq, bnr, (hipStream_t)stream);
    if (r != 0) return r < 0 ? r : 0;
  }
  {   // stride-1 (1,3,3) convs: the LDS-resident-patch kernel (csrc/conv_cl16_s3.hip)
    const int r = cl16_s3_try(g, mt, x, wl, y, in_ss, scale_shift, res, relu, stat_sum, stat_sq, bnr, (hipStream_t)stream);
    if (r != 0) return r < 0 ? r : 0;
  }
  const unsigned P = (unsigned)g.N * g.Lt * g.Lh * g.Lw;
  dim3 grid((P + CL_BN - 1) / CL_BN, g.Mrows / (16 * mt));
  const int pro = in_ss ? 1 : 0, epi = bnr.part ? 2 : (stat_sum ? 1 : 0);
#define SLV_CL16(MT_, PRO_, EPI_)                                                                                     \
  hipLaunchKernelGGL((conv_cl16_kernel<MT_, PRO_, EPI_>), grid, dim3(256), 0, (hipStream_t)stream,                    \
                     (const unsigned short*)x, (const unsigned short*)wl, (unsigned short*)y, in_ss, scale_shift,     \
                     (const unsigned short*)res, relu, stat_sum, stat_sq, bnr, g, make_fastdiv(g.Lw), make_fastdiv(g.Lh), \
                     make_fastdiv(g.Lt), make_fastdiv(grid.x))
#define SLV_CL16_MT(MT_)                              \
  do {                                                \
    if (epi == 2) SLV_CL16(MT_, 0, 2);                \
    else if (pro == 0 && epi == 0) SLV_CL16(MT_, 0, 0); \
    else if (pro == 1 && epi == 0) SLV_CL16(MT_, 1, 0); \
    else if (pro == 0 && epi == 1) SLV_CL16(MT_, 0, 1); \
    else SLV_CL16(MT_, 1, 1);                          \
  } while (0)
  if (mt == 4) SLV_CL16_MT(4);
  else if (mt == 8) SLV_CL16_MT(8);
  else SLV_CL16_MT(9);
#undef SLV_CL16_MT
#undef SLV_CL16
  return ::slv::launch_check(fn);
}

// geom: 20 int32 = {N, Ti, Hi, Wi, Cin_p, Cout, Cout_p, To, Ho, Wo, kt, kh, kw, st, sh, sw, pt, ph, pw, Mrows}.
// MT (16-row tiles per block) follows from geom.Mrows: the Python side picks it.
int slv_conv_cl16_fwd(const int32_t* geom, int mt, const void* x_bf16, const void* w_layout_bf16, void* y_bf16,
                      const float* scale_shift, const void* res_bf16, int relu, slv_stream_t stream) {
  using namespace slv;
  SLV_CHECK_ARG(geom && x_bf16 && w_layout_bf16 && y_bf16, "null pointer");
  const int32_t* q = geom;
  const int N = q[0], Ti = q[1], Hi = q[2], Wi = q[3], Cin_p = q[4], Cout = q[5], Cout_p = q[6], To = q[7], Ho = q[8],
            Wo = q[9], kt = q[10], kh = q[11], kw = q[12], st = q[13], sh = q[14], sw = q[15], pt = q[16], ph = q[17],
            pw = q[18], Mrows = q[19];
  SLV_CHECK_ARG(N > 0 && Cin_p > 0 && (Cin_p & 31) == 0 && Cout > 0 && Cout_p >= Cout && (Cout_p & 7) == 0,
                "channel counts (Cin_p % 32, Cout_p % 8)");
  SLV_CHECK_ARG(kt > 0 && kh > 0 && kw > 0 && st > 0 && sh > 0 && sw > 0 && kt <= 8 && kh <= 8 && kw <= 8 &&
                    kt * kh * kw <= 64, "kernel / stride");
  SLV_CHECK_ARG(To == (Ti + 2 * pt - kt) / st + 1 && Ho == (Hi + 2 * ph - kh) / sh + 1 &&
                    Wo == (Wi + 2 * pw - kw) / sw + 1 && To > 0 && Ho > 0 && Wo > 0,
                "output extent does not match the geometry");
  SLV_CHECK_ARG((long long)N * Ti * Hi * Wi * Cin_p * 2 < 0xFFFFFFF0LL && (long long)N * To * Ho * Wo * Cout_p * 2 < 0xFFFFFFF0LL,
                "tensor beyond the 32-bit buffer range");
  SLV_CHECK_ARG(mt == 4 || mt == 8 || mt == 9, "M tile (4, 8 or 9 x 16 rows)");
  SLV_CHECK_ARG(Mrows % (16 * mt) == 0 && Mrows >= Cout, "weight layout rows");
  ClConv g;
  memset(&g, 0, sizeof(g));
  g.N = N; g.Ti = Ti; g.Hi = Hi; g.Wi = Wi; g.Cin_p = Cin_p; g.Cin = Cin_p;
  g.Lt = To; g.Lh = Ho; g.Lw = Wo;
  g.bmt = st; g.bmh = sh; g.bmw = sw; g.bot = -pt; g.boh = -ph; g.bow = -pw;
  g.To = To; g.Ho = Ho; g.Wo = Wo; g.Cout = Cout; g.Cout_p = Cout_p;
  g.omt = g.omh = g.omw = 1;
  g.Mrows = Mrows;
  g.flags = 1;
  for (int a = 0; a < kt; ++a)
    for (int b = 0; b < kh; ++b)
      for (int c = 0; c < kw; ++c) g.tap[g.ntaps] = (a + 8) | (b + 8) << 4 | (c + 8) << 8 | g.ntaps << 12, ++g.ntaps;
  return cl16_launch(g, mt, x_bf16, w_layout_bf16, y_bf16, nullptr, scale_shift, res_bf16, relu, nullptr, nullptr,
                     ClBnr{nullptr, nullptr, nullptr, nullptr, 0, 0}, stream, __func__);
}

int32_t slv_cl16_conv_words(void) { return slv::CLC_WORDS; }
int32_t slv_cl16_g8_mode(int32_t mode) { return slv::cl16_g8_set_mode(mode); }
int32_t slv_cl16_conv_nblk(const int32_t* clconv);

int32_t slv_cl16_conv_dgrad_bn_apply_ok(const int32_t* clconv) {
  if (!clconv) return 0;
  slv::ClConv g;
  memcpy(&g, clconv, sizeof(g));
  return slv::cl16_tr_dgrad_apply_ok(g) ? 1 : 0;
}

int slv_cl16_conv_dgrad_bn_apply(const int32_t* clconv, const void* dy_bf16, const void* w_layout_bf16, void* out_bf16,
                                 const void* src_x_bf16, const float* bwd5, slv_stream_t stream) {
  using namespace slv;
  SLV_CHECK_ARG(clconv && dy_bf16 && w_layout_bf16 && out_bf16 && src_x_bf16 && bwd5, "null pointer");
  ClConv g;
  memcpy(&g, clconv, sizeof(g));
  SLV_CHECK_ARG(cl16_tr_dgrad_apply_ok(g), "not a launch this path takes (slv_cl16_conv_dgrad_bn_apply_ok)");
  const int rc = cl16_tr_dgrad_apply(g, dy_bf16, w_layout_bf16, out_bf16, src_x_bf16, bwd5, (hipStream_t)stream);
  return rc == 1 ? 0 : (rc ? rc : -1);
}

int slv_cl16_conv(const int32_t* clconv, int mt, const void* x_bf16, const void* w_layout_bf16, void* y_bf16,
                  const float* in_scale_shift, const float* scale_shift, const void* res_bf16, int relu,
                  float* stat_sum, float* stat_sq, const void* bnr_x_bf16, const float* bnr_scale_shift,
                  const float* bnr_mean_invstd, float* bnr_part, int bnr_slot0, int bnr_nslots, slv_stream_t stream) {
  using namespace slv;
  SLV_CHECK_ARG(clconv && x_bf16 && w_layout_bf16 && y_bf16, "null pointer");
  ClConv g;
  memcpy(&g, clconv, sizeof(g));
  SLV_CHECK_ARG(g.N > 0 && g.Cin_p > 0 && (g.Cin_p & 31) == 0 && g.Cin > 0 && g.Cin <= g.Cin_p && g.Cout > 0 &&
                    g.Cout_p >= g.Cout && (g.Cout_p & 7) == 0, "channel counts (Cin_p % 32, Cout_p % 8)");
  SLV_CHECK_ARG(g.Lt > 0 && g.Lh > 0 && g.Lw > 0 && g.Ti > 0 && g.Hi > 0 && g.Wi > 0 && g.To > 0 && g.Ho > 0 && g.Wo > 0,
                "extents");
  SLV_CHECK_ARG(g.ntaps >= 0 && g.ntaps <= 64, "taps");
  SLV_CHECK_ARG((g.Lt - 1) * g.omt + g.oot < g.To && (g.Lh - 1) * g.omh + g.ooh < g.Ho && (g.Lw - 1) * g.omw + g.oow < g.Wo &&
                    g.oot >= 0 && g.ooh >= 0 && g.oow >= 0 && g.omt > 0 && g.omh > 0 && g.omw > 0,
                "the lattice does not fit the output tensor");
  SLV_CHECK_ARG((long long)g.N * g.Ti * g.Hi * g.Wi * g.Cin_p * 2 < 0xFFFFFFF0LL &&
                    (long long)g.N * g.To * g.Ho * g.Wo * g.Cout_p * 2 < 0xFFFFFFF0LL &&
                    (long long)g.N * g.Lt * g.Lh * g.Lw < 0xFFFFFF00LL, "tensor beyond the 32-bit buffer range");
  SLV_CHECK_ARG(mt == 4 || mt == 8 || mt == 9, "M tile (4, 8 or 9 x 16 rows)");
  SLV_CHECK_ARG(g.Mrows % (16 * mt) == 0 && g.Mrows >= g.Cout, "weight layout rows");
  SLV_CHECK_ARG(!in_scale_shift || g.Cin_p <= CL_PRO_MAXC, "prologue table: Cin_p <= 1152");
  SLV_CHECK_ARG((stat_sum == nullptr) == (stat_sq == nullptr), "stat_sum / stat_sq come together");
  SLV_CHECK_ARG(!stat_sum || (!scale_shift && !res_bf16), "the statistics epilogue stores the raw output");
  SLV_CHECK_ARG(!bnr_part || (bnr_x_bf16 && bnr_scale_shift && bnr_mean_invstd && bnr_slot0 >= 0 &&
                              bnr_slot0 + slv_cl16_conv_nblk(clconv) <= bnr_nslots),
                "fused BatchNorm-backward sums: x, scale_shift, mean_invstd, slot range");
  SLV_CHECK_ARG(!bnr_part || (!in_scale_shift && !stat_sum && !scale_shift && !relu),
                "the fused sums belong to a plain backward-data launch");
  return cl16_launch(g, mt, x_bf16, w_layout_bf16, y_bf16, in_scale_shift, scale_shift, res_bf16, relu, stat_sum, stat_sq,
                     ClBnr{(const unsigned short*)bnr_x_bf16, bnr_scale_shift, bnr_mean_invstd, bnr_part, bnr_slot0, bnr_nslots},
                     stream, __func__);
}

int32_t slv_cl16_conv_nblk(const int32_t* clconv) {
  slv::ClConv g;
  memcpy(&g, clconv, sizeof(g));
  if (slv::cl16_sr_applies(g)) return slv::cl16_sr_slots(g);      // one partial per persistent workgroup
  if (slv::cl16_tr_applies(g) && slv::cl16_tr_forward(g)) return slv::cl16_tr_columns(g);      // one partial per 32-pixel column
  const long long P = (long long)g.N * g.Lt * g.Lh * g.Lw;
  // positions per tile of the kernel that runs (the order cl16_launch tries them in)
  const int bn = slv::cl16_g8_applies(g) ? slv::cl16_g8_positions() : slv::cl16_s3_applies(g) ? slv::cl16_s3_positions() : slv::CL_BN;
  return (int32_t)((P + bn - 1) / bn);
}

int slv_to_cl16(const float* x, void* y_bf16, int64_t N, int C, int Cp, int64_t S, slv_stream_t stream) {
  using namespace slv;
  SLV_CHECK_ARG(x && y_bf16 && N > 0 && C > 0 && Cp >= C && (Cp & 7) == 0 && S > 0, "bad argument");
  const long long total = N * S * (Cp / 8);
  SLV_CHECK_ARG(total < 0xFFFFFFFFLL, "tensor too large");
  hipLaunchKernelGGL(to_cl16_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x,
                     (unsigned short*)y_bf16, C, Cp, (unsigned)S, (unsigned)total);
  SLV_LAUNCH_CHECK();
  return 0;
}

int slv_maxpool_cl16(const void* x_bf16, void* y_bf16, int64_t N, int H, int W, int Cp, slv_stream_t stream) {
  using namespace slv;
  SLV_CHECK_ARG(x_bf16 && y_bf16 && N > 0 && H > 0 && W > 0 && Cp > 0 && (Cp & 7) == 0, "bad argument");
  const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
  const long long total = N * Ho * Wo * (Cp / 8);
  SLV_CHECK_ARG(total < 0xFFFFFFFFLL, "tensor too large");
  hipLaunchKernelGGL(maxpool_cl16_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     (const unsigned short*)x_bf16, (unsigned short*)y_bf16, H, W, Ho, Wo, Cp, (unsigned)total);
  SLV_LAUNCH_CHECK();
  return 0;
}

int slv_avgpool_cl16(const void* x_bf16, float* y, int64_t N, int64_t S, int C, int Cp, slv_stream_t stream) {
  using namespace slv;
  SLV_CHECK_ARG(x_bf16 && y && N > 0 && N <= 65535 && S > 0 && C > 0 && Cp >= C, "bad argument");
  hipLaunchKernelGGL(avgpool_cl16_kernel, dim3((C + 63) / 64, (unsigned)N), dim3(64), 0, (hipStream_t)stream,
                     (const unsigned short*)x_bf16, y, (int)S, C, Cp);
  SLV_LAUNCH_CHECK();
  return 0;
}

int slv_to_cl16_wpatch(const float* x, void* y_bf16, int64_t N, int C, int64_t TH, int W, int kw, int sw, int pw,
                       slv_stream_t stream) {
  using namespace slv;
  SLV_CHECK_ARG(x && y_bf16 && N > 0 && C > 0 && TH > 0 && W > 0 && kw > 0 && sw > 0 && kw * C <= 32,
                "bad argument (kw * C must fit 32)");
  const int Wo = (W + 2 * pw - kw) / sw + 1;
  const long long total = N * TH * Wo * 4;
  SLV_CHECK_ARG(Wo > 0 && total < 0xFFFFFFFFLL && TH < 0x7FFFFFFFLL, "tensor too large");
  hipLaunchKernelGGL(to_cl16_wpatch_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x,
                     (unsigned short*)y_bf16, C, (unsigned)TH, W, Wo, kw, sw, pw, (unsigned)total);
  SLV_LAUNCH_CHECK();
  return 0;
}

}  // extern "C"
