// Weight gradient of the stride-1 (1,3,3) convs of the 16-bit path (BASELINE configs[4]; the spatial convs of R(2+1)D:
// /root/reference/model.py:147-176 builds torchvision's r2plus1d_18, main.py:296-299 runs its backward) with the
// activation rows staged ONCE for the three taps of a kernel row.
//
// The general kernel (train_cl16.hip: cl16_wgrad_kernel) treats every (tap, channel chunk) column block on its own: the
// position decode, the bounds checks and the BatchNorm + ReLU prologue of an activation piece are repeated for each of
// the 9 taps -- 116 VALU instructions per 15 MFMAs in its K loop (7.7 : 1; the loop is VALU-bound).  Here a block owns
//   (Cout tile of 32*WM rows) x (ONE kernel row eh, its 3 taps ew = -1, 0, 1) x (one 64-channel group) x (a K slice)
// and keeps a rolling PATCH of activation rows in LDS: a K step of 32 positions adds 32 new rows (one 16-byte piece per
// thread: decoded, checked and activated once) and the three taps read them at row offsets -1, 0, +1.
//   * Padded row index: position q sits at patch row q + q / W, i.e. after every image row comes one ZERO row, so a tap
//     that leaves the image to the left or right lands on zeros and a tap offset is a constant row offset
//     eh * (W + 1) + ew: no per-tap masks.  (The index is taken relative to an origin some rows before the slice.)
//   * Leaving the image at the top / bottom means reading the neighbouring frame's last / first row: with eh fixed per
//     block those rows (h == H-1 for eh = -1, h == 0 for eh = +1) are zeroed when they are staged.
//   * The patch is circular (128 rows of 160 bytes: three staging units of <= 40 rows are live), dY tiles are double
//     buffered and arrive by LDS-DMA (buffer_load ... lds; two padding pieces per row keep the rows 32 bytes (mod 64)
//     apart for the transpose reads), fragments come from ds_read_b64_tr_b16 exactly as in the general kernel.
// Output: the same deterministic split-K partials [slice][rows][9 * Cin_p] that cl16_wgrad_reduce_kernel sums.
#include "cl16.hpp"

namespace slv {

#ifndef SLV_WG3_ABL
#define SLV_WG3_ABL 0            // timing ablations (tools/build_variant.sh): 1 no MFMA, 2 no dY loads, 3 no activation loads
#endif
constexpr int W3_SP = 160, W3_PR = 128;          // patch row bytes (128 + 32), circular patch rows

template <int WM, int PRO>
__global__ __launch_bounds__(256, 2) void cl16_wgrad3_kernel(const unsigned short* __restrict__ dy,
                                                             const unsigned short* __restrict__ x,
                                                             const float* __restrict__ in_ss, float* __restrict__ part,
                                                             ClWgrad3 g, FastDiv dW, FastDiv dH) {
  constexpr int BM = 32 * WM, APC = BM / 8 + 2, SA = APC * 16, AIT = (32 * APC + 255) / 256;
  constexpr int ABYTES = AIT * 4096;                              // one dY buffer (the DMA writes whole 1 KiB wave pieces)
  constexpr int NAB = 3;                                           // dY buffers: the tile of step s + 2 is in flight during step s
  // Two LDS objects and hand-issued reads of the first: the compiler orders every LDS read it can see behind LDS-DMA
  // "that may alias" with s_waitcnt vmcnt(0) -- in front of the fragment reads of a step that is the whole memory latency
  // of the requests the step has just issued (measured: 1 850 of a step's 3 070 cycles in "fragment reads + 30 MFMA").
  // The dY buffers (the DMA's only target) are read with inline-asm ds_read_b64_tr_b16; the patch is a separate array
  // that no DMA writes.  The real dependences are ordered by the s_waitcnt vmcnt + barrier at the end of a step.
  __shared__ __attribute__((aligned(16))) unsigned char lds_a[NAB * ABYTES];
  __shared__ __attribute__((aligned(16))) unsigned char patch[W3_PR * W3_SP];
  unsigned char* const lds_raw = lds_a;
  typedef __attribute__((address_space(3))) void* lds_void;
  typedef __attribute__((address_space(3))) s16x4* lds_s16x4;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
  // XCD-aware bijective remap: consecutive units (same K slice: same dY and activation rows) share an XCD's L2
  const unsigned total = gridDim.x, q8 = total >> 3, r8 = total & 7, xcd = blockIdx.x & 7, loc = blockIdx.x >> 3;
  unsigned unit = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + loc;
  const int ehi = unit % 3; unit /= 3;
  const int grp = unit % g.groups; unit /= g.groups;
  const int mt = unit % g.mtiles;
  const unsigned slice = unit / g.mtiles;
  const int eh = ehi - 1, m0 = mt * BM, c0 = grp * 64;
  const int W = g.W, H = g.H;
  const unsigned P = (unsigned)g.N * g.T * g.H * g.W;
  const unsigned k_lo = slice * (unsigned)g.kper, k_hi = min(k_lo + (unsigned)g.kper, P);
  const int nsteps = k_lo < k_hi ? (int)((k_hi - k_lo + 31) >> 5) : 0;
  // origin of the relative row numbering: a whole number of image rows, far enough before the slice's first staged row
  const int base_row = (int)fdiv(k_lo, dW) - (3 + 32 / W);
  const int org = base_row * W;                                      // may be negative
  const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc((void*)dy, 0, (int)(P * (unsigned)g.Cout_p * 2u), 0x00020000);
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, (int)(P * (unsigned)g.Cin_p * 2u), 0x00020000);

  // ---- dY tile by LDS-DMA: piece pc = tid + 256 i -> LDS byte pc * 16 = row (pc / APC) * SA + column piece * 16
  unsigned avo[AIT];
#pragma unroll
  for (int i = 0; i < AIT; ++i) {
    const int pc = tid + 256 * i, row = pc / APC, col = pc - row * APC, cch = m0 + col * 8;
    avo[i] = (col < APC - 2 && cch < g.Cout_p) ? (unsigned)(row * g.Cout_p * 2 + cch * 2) : 0xFFFFFFFFu;
  }
  // 32 * APC is a multiple of 64: a wave's piece is inside the tile or entirely outside (only the last i can be): a wave
  // issues AIT or AIT - 1 operations per tile, and the step's s_waitcnt counts with the wave's own number
  constexpr bool PARTIAL = (32 * APC) % 256 != 0;
  const bool last_in = !PARTIAL || wave * 64 + 256 * (AIT - 1) < 32 * APC;                   // wave-uniform
  auto dma_a = [&](unsigned k0, int buf) __attribute__((always_inline)) {
    const unsigned kb = k0 * (unsigned)(g.Cout_p * 2);
    unsigned char* dst = lds_raw + buf * ABYTES + wave * 1024;
#pragma unroll
    for (int i = 0; i < AIT; ++i) {
      if (SLV_WG3_ABL == 2) continue;
      if (i < AIT - 1 || last_in)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(ry, (lds_void)(dst + i * 4096), 16,
                                                 (int)(avo[i] == 0xFFFFFFFFu ? 0xFFFFFFF0u : avo[i] + kb), 0, 0, 0);
    }
  };

  // ---- activation rows: staging unit u = the 32 rows q = k_lo + eh * W + 1 + 32 (u - 1) + j, one piece per thread
  const int sj = tid >> 3, sc8 = tid & 7, scx = c0 + sc8 * 8;
  const bool scv = scx < g.Cin_p;
  float ps[8], ph[8];
  if constexpr (PRO == 1) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const bool ok = scx + i < g.Cin;
      ps[i] = ok ? in_ss[scx + i] : 0.f;
      ph[i] = ok ? in_ss[g.Cin + scx + i] : 0.f;
    }
  }
  struct Staged {
    u32x4 reg;
    int slot;
    bool ok, pad;
  };
  Staged sg[2];                                                     // two units in flight (indexed with compile-time parities only)
  auto stage_load = [&](int u, Staged& S) __attribute__((always_inline)) {
    const int q = (int)k_lo + eh * W + 1 + 32 * (u - 1) + sj;
    bool ok = (unsigned)q < P && scv;
    const unsigned qr = (unsigned)(q - org), rr = fdiv(qr, dW), wq = qr - rr * (unsigned)W;
    if (eh != 0) {                                                   // the row above / below the image: zeros
      const unsigned ar = ok ? (unsigned)((int)rr + base_row) : 0u;
      const unsigned hq = ar - fdiv(ar, dH) * (unsigned)H;
      if (hq == (eh > 0 ? 0u : (unsigned)(H - 1))) ok = false;
    }
    if (SLV_WG3_ABL == 3) S.reg = (u32x4){(unsigned)q, 1u, 2u, 3u};
    else S.reg = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(
                                               rx, ok ? (unsigned)q * (unsigned)(g.Cin_p * 2) + (unsigned)scx * 2u : 0xFFFFFFF0u, 0, 0));
    S.slot = (int)((qr + rr) & (W3_PR - 1));
    S.ok = ok;
    S.pad = wq == (unsigned)(W - 1);
  };
  auto stage_store = [&](const Staged& S) __attribute__((always_inline)) {
    u32x4 v = S.reg;
    if constexpr (PRO == 1) {
      const u32x4 t = affine_relu8(v, ps, ph);
      v = S.ok ? t : (u32x4){0u, 0u, 0u, 0u};
    }
    *(u32x4*)(patch + S.slot * W3_SP + sc8 * 16) = v;
    if (S.pad) *(u32x4*)(patch + ((S.slot + 1) & (W3_PR - 1)) * W3_SP + sc8 * 16) = (u32x4){0u, 0u, 0u, 0u};
  };

  f32x4 acc[WM][3][2];
#pragma unroll
  for (int i = 0; i < WM; ++i)
#pragma unroll
    for (int e = 0; e < 3; ++e)
#pragma unroll
      for (int c = 0; c < 2; ++c) acc[i][e][c] = (f32x4){0.f, 0.f, 0.f, 0.f};
  // transpose-read lanes (cf. cl16_wgrad_kernel): tile rows 4 fg + (fi >> 2) (+16), 4 columns at 4 (fi & 3)
  const int fg = lane >> 4, fi = lane & 15;
  const int rlo = 4 * fg + (fi >> 2);
  const int fa = rlo * SA + (wm * WM * 16 + 4 * (fi & 3)) * 2;
  const int fb = (wn * 32 + 4 * (fi & 3)) * 2;
  const unsigned a_lds = (unsigned)(unsigned long long)(__attribute__((address_space(3))) unsigned char*)lds_a;   // LDS byte address

  // Pipeline: at the top of step s the dY tile of step s + 2 (three buffers, LDS-DMA) and the activation rows of unit
  // s + 3 (registers; written to the patch at the end of step s + 1) are requested: both have more than a full step to
  // arrive.  Vector memory operations complete in order and a wave issues a fixed number of them per step (1 + its DMA
  // operations), so "at most that many outstanding" at the end of step s means the dY tile of step s + 1 has landed.
#ifdef SLV_WG3_TRACE   // s_memtime at 6 points of steps 8..15 of blocks 0..31 (wave 0) -> the first 32 * 64 uint64 of `part` (timing only!)
#define W3_T(slot) if (tr_on && s >= 8 && s < 16) trace[(s - 8) * 6 + (slot)] = __builtin_amdgcn_s_memtime()
  const bool tr_on = tid == 0 && blockIdx.x < 32;
  unsigned long long* trace = (unsigned long long*)part + (size_t)blockIdx.x * 64;
  if (tr_on) trace[60] = 0x1234u + (unsigned long long)nsteps;
#else
#define W3_T(slot)
#endif
  auto step = [&](int s, int ab, Staged& Snew, const Staged& Sold) __attribute__((always_inline)) {
    const unsigned k0 = k_lo + 32u * (unsigned)s;
    const int ab2 = ab == 0 ? 2 : ab - 1;                            // (s + 2) % 3
    W3_T(0);
    stage_load(s + 3, Snew);
    W3_T(1);
    dma_a(k0 + 64u, ab2);                                            // past the slice: rows nobody reads (past P: zeros)
    W3_T(2);
    {
      // patch rows of this lane's positions k0 + rlo and k0 + rlo + 16 for the tap row, column offset included
      unsigned prow[2];
#pragma unroll
      for (int hl = 0; hl < 2; ++hl) {
        const unsigned pr = k0 + (unsigned)(rlo + 16 * hl) - (unsigned)org;
        prow[hl] = pr + fdiv(pr, dW) + (unsigned)(eh * (W + 1));
      }
      bf16x8 a[WM], b[3][2];
      u32x2 alo[WM], ahi[WM];
      const unsigned aaddr = a_lds + (unsigned)(ab * ABYTES + fa);
#pragma unroll
      for (int i = 0; i < WM; ++i) {
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(alo[i]) : "v"(aaddr), "n"(i * 32) : "memory");
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(ahi[i]) : "v"(aaddr), "n"(i * 32 + 16 * SA) : "memory");
      }
#pragma unroll
      for (int e = 0; e < 3; ++e) {
        const unsigned char* blo = patch + ((prow[0] + (unsigned)(e - 1)) & (W3_PR - 1)) * W3_SP + fb;
        const unsigned char* bhi = patch + ((prow[1] + (unsigned)(e - 1)) & (W3_PR - 1)) * W3_SP + fb;
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(blo + c * 32));
          const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(bhi + c * 32));
          b[e][c] = tr_pair(lo, hi);
        }
      }
      // The hand-issued reads are invisible to the compiler's counters, but every MFMA also takes a patch fragment, and
      // those reads are the compiler's own, issued AFTER the hand-issued ones (the asm's "memory" clobber keeps them
      // there): LDS operations return in order, so the compiler's wait for b[e][c] covers every a[i].  Tap-major MFMA
      // order: the first MFMAs start when the first patch fragment is in, not after all 22 reads.
#pragma unroll
      for (int i = 0; i < WM; ++i) {
        const u32x4 t = {alo[i][0], alo[i][1], ahi[i][0], ahi[i][1]};
        a[i] = __builtin_bit_cast(bf16x8, t);
      }
#pragma unroll
      for (int e = 0; e < 3; ++e)
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
          for (int i = 0; i < WM; ++i) {
            if (SLV_WG3_ABL == 1) acc[i][e][c][0] += __builtin_bit_cast(float, __builtin_bit_cast(u32x4, a[i])[0] ^ __builtin_bit_cast(u32x4, b[e][c])[0]);
            else acc[i][e][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[e][c], acc[i][e][c], 0, 0, 0);
          }
      __builtin_amdgcn_sched_barrier(0);
    }
    W3_T(3);
    stage_store(Sold);                                               // unit s + 2, requested one step ago
    W3_T(4);
    // (not __syncthreads(): its release fence makes the compiler wait for ALL vector memory operations, the DMA of step
    //  s + 2 included; lgkmcnt(0): this wave's patch writes are done before the barrier publishes them)
    if (last_in) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(SLV_WG3_ABL == 2 ? 1 : 1 + AIT) : "memory");
    else asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(SLV_WG3_ABL == 2 ? 1 : AIT) : "memory");
    W3_T(5);
  };
  if (nsteps > 0) {
    stage_load(0, sg[0]);
    dma_a(k_lo, 0);
    dma_a(k_lo + 32u, 1);
    stage_store(sg[0]);
    stage_load(1, sg[0]);
    stage_store(sg[0]);
    stage_load(2, sg[1]);                                            // stored at the end of step 0
    asm volatile("s_waitcnt vmcnt(1)" ::: "memory");               // everything but that last request
    __syncthreads();
  }
  for (int s = 0; s < nsteps; s += 2) {                              // s % 3 cycles 0, 2, 1 over the pairs
    const int ab = s % 3;
    step(s, ab, sg[0], sg[1]);
    if (s + 1 < nsteps) step(s + 1, ab == 2 ? 0 : ab + 1, sg[1], sg[0]);
  }
  // ---- partial tile of this K slice: part[slice][mtiles * BM][9 * Cin_p]; C/D: col = lane & 15, rows (lane >> 4) * 4 + r
  const size_t ldp = (size_t)9 * g.Cin_p;
  float* pt = part + ((size_t)slice * g.mtiles * BM + m0 + wm * WM * 16) * ldp;
#ifdef SLV_WG3_TRACE
  if (acc[0][0][0][0] == 123.456f)
#endif
  if (c0 + wn * 32 < g.Cin_p) {
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
      for (int e = 0; e < 3; ++e)
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            pt[(size_t)(i * 16 + fg * 4 + r) * ldp + (size_t)(ehi * 3 + e) * g.Cin_p + c0 + wn * 32 + c * 16 + fi] = acc[i][e][c][r];
  }
}

// ---- host side ------------------------------------------------------------------------------------------------------
// Does the patch kernel take this weight gradient, and with which tiling?  (kt, kh, kw) = (1, 3, 3), stride 1, padding
// (0, 1, 1), W >= 4 (three staging units fit the circular patch).  SELAVI_CL16_WG3=0 sends everything to the general kernel.
bool wgrad3_plan(int N, int T, int H, int W, int Cin_p, int Cin, int Cout_p, int kt, int kh, int kw, int st, int sh, int sw,
                 int pt, int ph, int pw, int To, int Ho, int Wo, int* wm, ClWgrad3* out) {
  static int enabled = -1;
  if (enabled < 0) {
    const char* e = getenv("SELAVI_CL16_WG3");
    enabled = !(e && e[0] == '0');
  }
  if (!enabled) return false;
  if (kt != 1 || kh != 3 || kw != 3 || st != 1 || sh != 1 || sw != 1 || pt != 0 || ph != 1 || pw != 1) return false;
  if (To != T || Ho != H || Wo != W || W < 4) return false;
  int best = 5;
  long long best_rows = 1LL << 60;
  for (int w = 5; w >= 2; --w) {                                  // least padding, then the larger tile
    const long long rows = (long long)((Cout_p + 32 * w - 1) / (32 * w)) * 32 * w;
    if (rows < best_rows) {
      best_rows = rows;
      best = w;
    }
  }
  ClWgrad3 g;
  g.N = N; g.T = T; g.H = H; g.W = W; g.Cin_p = Cin_p; g.Cin = Cin; g.Cout_p = Cout_p;
  g.mtiles = (Cout_p + 32 * best - 1) / (32 * best);
  g.groups = (Cin_p + 63) / 64;
  const long long P = (long long)N * T * H * W;
  const long long base = (long long)g.mtiles * g.groups * 3;
  // ONE round of resident blocks: 2 per CU with WM = 3, 4, 5 (> 168 registers), 3 with WM = 2 (a grid of
  // 1.25 rounds runs at 62 %: measured with a target of 640 blocks on 512 slots); >= 64 K steps per slice
  const long long slots = best >= 3 ? 512 : 768;      // registers: 234 / 206 / 176 -> 2 blocks per CU, 148 (WM = 2) -> 3
  long long ksl = slots / base;
  if (ksl > P / 2048) ksl = P / 2048;
  if (ksl < 1) ksl = 1;
  long long kper = ((P + ksl - 1) / ksl + 31) / 32 * 32;
  g.kper = (int)kper;
  g.kslices = (int)((P + kper - 1) / kper);
  *wm = best;
  *out = g;
  return true;
}

size_t wgrad3_ws_bytes(const ClWgrad3& g, int wm) {
  return (size_t)g.kslices * g.mtiles * wm * 32 * 9 * g.Cin_p * sizeof(float);
}

template <int WM>
static void wgrad3_launch_wm(const ClWgrad3& g, const void* dy, const void* x, const float* in_ss, float* part, hipStream_t st) {
  const FastDiv dW = make_fastdiv(g.W), dH = make_fastdiv(g.H);
  const unsigned blocks = (unsigned)(g.kslices * g.mtiles * g.groups * 3);
  if (in_ss)
    hipLaunchKernelGGL((cl16_wgrad3_kernel<WM, 1>), dim3(blocks), dim3(256), 0, st, (const unsigned short*)dy,
                       (const unsigned short*)x, in_ss, part, g, dW, dH);
  else
    hipLaunchKernelGGL((cl16_wgrad3_kernel<WM, 0>), dim3(blocks), dim3(256), 0, st, (const unsigned short*)dy,
                       (const unsigned short*)x, in_ss, part, g, dW, dH);
}

void wgrad3_launch(const ClWgrad3& g, int wm, const void* dy, const void* x, const float* in_ss, float* part, hipStream_t st) {
  if (wm == 2) wgrad3_launch_wm<2>(g, dy, x, in_ss, part, st);
  else if (wm == 3) wgrad3_launch_wm<3>(g, dy, x, in_ss, part, st);
  else if (wm == 4) wgrad3_launch_wm<4>(g, dy, x, in_ss, part, st);
  else wgrad3_launch_wm<5>(g, dy, x, in_ss, part, st);
}

}  // namespace slv
