// Weight gradient of the stride-1 temporal (3,1,1) convs of the 16-bit path (BASELINE configs[4]; the second half of every
// Conv2Plus1D of R(2+1)D: /root/reference/model.py:147-176 builds torchvision's r2plus1d_18, main.py:296-299 runs its
// backward) with every activation row read from memory ONCE.
//
//   dW[dt][co][ci] = sum over positions (n, t, hw) of dY[n, t, hw][co] * act(X)[n, t + dt, hw][ci],   dt = -1, 0, 1
//
// The general kernel (train_cl16.hip) gives each tap its own column tile: the three taps' blocks read the activation
// rows of frames t-1, t, t+1 at the same moment, i.e. every row three times, H * W rows apart -- far beyond any cache
// (layer 1 at 64 clips: TCC hit rate 19 %, 4.3 GB through the fabric for 1.44 GB of tensors, 5.8 TB/s: the kernel sits
// on the HBM roofline of its own re-reads).  Here the contraction runs over positions in COLUMN order -- a block walks
// a 32-pixel column of a clip frame by frame -- and one block forms all three taps: the activation tile of frame t + 1
// is staged (decoded, bounds-checked, BatchNorm + ReLU applied: once) while frame t is multiplied, and a ring of four
// tiles holds frames t-1, t, t+1 for the taps -1, 0, +1 of dY's frame t.
//   * step s = column * (T + 1) + tv: after the T frames of a column comes one VIRTUAL frame of zeros (rows of frame
//     tv == T are never read from memory), so the tap that reaches before the first / past the last frame of a clip
//     lands on zeros and consecutive columns need no special case; its MFMAs are skipped.
//   * block = (Cout tile of 32 * WM rows) x (all three taps x one group of 32 * NC channels) x (a slice of steps);
//     the waves form a 2 x 2 grid, a wave owns WM x (3 taps x NC) 16 x 16 accumulator tiles ((WM, NC) = (2, 5) for the
//     64 x 160-channel layer-1 convs: the whole N extent in one block, (4, 2) / (5, 2) for the wider layers).
//   * dY tiles by LDS-DMA into three buffers, activation rows two steps ahead through registers, fragments by
//     ds_read_b64_tr_b16, hand-issued reads of the DMA buffers, barrier without the release fence: wgrad_cl16_s3.hip has
//     the why of each.
// Output: the deterministic split-K partials [slice][rows][3 * Cin_p] that cl16_wgrad_reduce_kernel sums.
#include "cl16.hpp"

namespace slv {

template <int WM, int NC, int PRO>
__global__ __launch_bounds__(256, 2) void cl16_wgrad_t_kernel(const unsigned short* __restrict__ dy,
                                                              const unsigned short* __restrict__ x,
                                                              const float* __restrict__ in_ss, float* __restrict__ part,
                                                              ClWgradT g, FastDiv dT1, FastDiv dPB, size_t kind_stride) {
  // PRO 2 (csrc/wgrad_cl16_t2.hip): consecutive units are the two KINDS of one (tile, slice) -- the activation operand is
  // the masked raw value m x (kind 0) or the mask m itself (kind 1), m = [x s + h > 0]; partials of kind k at k * kind_stride
  constexpr int BM = 32 * WM, APC = BM / 8 + 2, SA = APC * 16, AIT = (32 * APC + 255) / 256;
  constexpr int ABYTES = 32 * SA;             // pieces are linear (pc * 16) and a wave's piece outside the tile is skipped: no rounding to 4 KiB
  constexpr int NG = 32 * NC, NGP = NG / 8, SPB = NG * 2 + 32, TILEB = 32 * SPB, SIT = (32 * NGP + 255) / 256;
  constexpr int NAB = 3, NRING = 4;
  // two LDS objects: the DMA's target is read with hand-issued ds_read_b64_tr_b16 only (wgrad_cl16_s3.hip)
  __shared__ __attribute__((aligned(16))) unsigned char lds_a[NAB * ABYTES];
  __shared__ __attribute__((aligned(16))) unsigned char ring[NRING * TILEB + (PRO ? 2 * NG * 4 : 0)];
  static_assert(PRO >= 0 && PRO <= 2, "prologue kinds");
  float* const pro = (float*)(ring + NRING * TILEB);               // PRO 1: [2][NG] scale, shift of this group's channels
  typedef __attribute__((address_space(3))) void* lds_void;
  typedef __attribute__((address_space(3))) s16x4* lds_s16x4;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
  const unsigned total = gridDim.x, q8 = total >> 3, r8 = total & 7, xcd = blockIdx.x & 7, loc = blockIdx.x >> 3;
  unsigned unit = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + loc;
  const int kind = PRO == 2 ? (int)(unit & 1u) : 0;
  if constexpr (PRO == 2) unit >>= 1;
  const int grp = unit % g.groups; unit /= g.groups;
  const int mt = unit % g.mtiles;
  const unsigned slice = unit / g.mtiles;
  const int m0 = mt * BM, c0 = grp * NG;
  const int T1 = g.T + 1;
  const long long s_tot = (long long)g.N * g.PB * T1;              // steps of the whole tensor (< 2^31: the host checks)
  const int s_lo = (int)(slice * (unsigned)g.sper), s_hi = (int)min((long long)s_lo + g.sper, s_tot);
  const int nsteps = s_hi > s_lo ? s_hi - s_lo : 0;
  const unsigned Prows = (unsigned)g.N * g.T * g.HW;
  const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc((void*)dy, 0, (int)(Prows * (unsigned)g.Cout_p * 2u), 0x00020000);
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, (int)(Prows * (unsigned)g.Cin_p * 2u), 0x00020000);
  if constexpr (PRO != 0) {
    for (int i = tid; i < 2 * NG; i += 256) {
      const int c = c0 + (i % NG), which = i / NG;
      pro[i] = c < g.Cin ? in_ss[which * g.Cin + c] : 0.f;
    }
  }
  // step -> (first tensor row of its 32-pixel frame tile, rows that exist): -1 rows when the step is outside the tensor
  // or the virtual zero frame of its column
  auto step_rows = [&](int s, unsigned& row0) __attribute__((always_inline)) {
    if (s < 0 || (long long)s >= s_tot) return 0;
    const unsigned col = fdiv((unsigned)s, dT1), tv = (unsigned)s - col * (unsigned)T1;
    if ((int)tv >= g.T) return 0;
    const unsigned n = fdiv(col, dPB), pb = col - n * (unsigned)g.PB, px0 = pb * 32u;
    row0 = (n * (unsigned)g.T + tv) * (unsigned)g.HW + px0;
    return min(32, g.HW - (int)px0);
  };

  // ---- dY tile by LDS-DMA (cf. wgrad_cl16_s3.hip): piece pc = tid + 256 i -> row pc / APC, column piece pc % APC
  unsigned avo[AIT];
  int arow[AIT];
#pragma unroll
  for (int i = 0; i < AIT; ++i) {
    const int pc = tid + 256 * i, row = pc / APC, col = pc - row * APC, cch = m0 + col * 8;
    arow[i] = row;
    avo[i] = (col < APC - 2 && cch < g.Cout_p) ? (unsigned)(row * g.Cout_p * 2 + cch * 2) : 0xFFFFFFFFu;
  }
  constexpr bool PARTIAL = (32 * APC) % 256 != 0;
  const bool last_in = !PARTIAL || wave * 64 + 256 * (AIT - 1) < 32 * APC;                   // wave-uniform
  auto dma_a = [&](int s, int buf) __attribute__((always_inline)) {
    unsigned row0 = 0;
    const int nr = step_rows(s, row0);
    const unsigned kb = row0 * (unsigned)(g.Cout_p * 2);
    unsigned char* dst = lds_a + buf * ABYTES + wave * 1024;
#pragma unroll
    for (int i = 0; i < AIT; ++i)
      if (i < AIT - 1 || last_in)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(ry, (lds_void)(dst + i * 4096), 16,
                                                 (int)((avo[i] == 0xFFFFFFFFu || arow[i] >= nr) ? 0xFFFFFFF0u : avo[i] + kb), 0, 0, 0);
  };

  // ---- activation tile of a step: 32 rows x NGP pieces, SIT per thread, through registers (prologue math at the store)
  struct Staged {
    u32x4 reg[SIT];
    unsigned okm;                                                   // bit i: piece i is inside the tensor
  };
  Staged sg[2];
  auto stage_load = [&](int s, Staged& S) __attribute__((always_inline)) {
    unsigned row0 = 0;
    const int nr = step_rows(s, row0);
    S.okm = 0;
#pragma unroll
    for (int i = 0; i < SIT; ++i) {
      const int idx = tid + 256 * i, j = idx / NGP, pc = idx - j * NGP, ch = c0 + pc * 8;
      const bool ok = idx < 32 * NGP && j < nr && ch < g.Cin_p;
      S.okm |= (unsigned)ok << i;
      S.reg[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(
                                               rx, ok ? (row0 + (unsigned)j) * (unsigned)(g.Cin_p * 2) + (unsigned)ch * 2u : 0xFFFFFFF0u, 0, 0));
    }
  };
  auto stage_store = [&](int s, const Staged& S) __attribute__((always_inline)) {
    unsigned char* dst = ring + (s & (NRING - 1)) * TILEB;
#pragma unroll
    for (int i = 0; i < SIT; ++i) {
      const int idx = tid + 256 * i, j = idx / NGP, pc = idx - j * NGP;
      if (idx < 32 * NGP) {
        u32x4 v = S.reg[i];
        if constexpr (PRO == 1) {
          float sc[8], sh[8];
          const float* sp = pro + pc * 8;
          *(f32x4*)sc = *(const f32x4*)sp;
          *(f32x4*)(sc + 4) = *(const f32x4*)(sp + 4);
          *(f32x4*)sh = *(const f32x4*)(sp + NG);
          *(f32x4*)(sh + 4) = *(const f32x4*)(sp + NG + 4);
          const u32x4 t = affine_relu8(v, sc, sh);
          v = ((S.okm >> i) & 1) ? t : (u32x4){0u, 0u, 0u, 0u};
        }
        if constexpr (PRO == 2) {
          const float* sp = pro + pc * 8;
          const bool live = (S.okm >> i) & 1;
          const u32x4 base = kind ? (u32x4){0x3F803F80u, 0x3F803F80u, 0x3F803F80u, 0x3F803F80u} : v;      // bf16 1.0 / x
#pragma unroll
          for (int d = 0; d < 4; ++d) {
            // the mask as every kernel of this path evaluates it: one fused multiply-add in fp32, > 0
            const bool lo = live && bn_affine(bf_lo(v[d]), sp[2 * d], sp[NG + 2 * d]) > 0.f;
            const bool hi = live && bn_affine(bf_hi(v[d]), sp[2 * d + 1], sp[NG + 2 * d + 1]) > 0.f;
            v[d] = base[d] & ((lo ? 0x0000FFFFu : 0u) | (hi ? 0xFFFF0000u : 0u));
          }
        }
        *(u32x4*)(dst + j * SPB + pc * 16) = v;
      }
    }
  };

  f32x4 acc[WM][3][NC];
#pragma unroll
  for (int i = 0; i < WM; ++i)
#pragma unroll
    for (int e = 0; e < 3; ++e)
#pragma unroll
      for (int c = 0; c < NC; ++c) acc[i][e][c] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const int fg = lane >> 4, fi = lane & 15;
  const int rlo = 4 * fg + (fi >> 2);
  const int fa = rlo * SA + (wm * WM * 16 + 4 * (fi & 3)) * 2;
  const int fb = rlo * SPB + (wn * NC * 16 + 4 * (fi & 3)) * 2;
  const unsigned a_lds = (unsigned)(unsigned long long)(__attribute__((address_space(3))) unsigned char*)lds_a;

  // step s (global index), dY buffer ab = (s - s_lo) % 3; Snew takes tile s + 3, Sold holds tile s + 2
  auto step = [&](int s, int ab, Staged& Snew, const Staged& Sold) __attribute__((always_inline)) {
    const int ab2 = ab == 0 ? 2 : ab - 1;                            // buffer of step s + 2
    stage_load(s + 3, Snew);
    dma_a(s + 2, ab2);
    unsigned row0 = 0;
    if (step_rows(s, row0) > 0) {                                    // (the virtual zero frame: nothing to multiply)
      u32x2 alo[WM], ahi[WM];
      const unsigned aaddr = a_lds + (unsigned)(ab * ABYTES + fa);
#pragma unroll
      for (int i = 0; i < WM; ++i) {
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(alo[i]) : "v"(aaddr), "n"(i * 32) : "memory");
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(ahi[i]) : "v"(aaddr), "n"(i * 32 + 16 * SA) : "memory");
      }
      bf16x8 a[WM];
#pragma unroll
      for (int i = 0; i < WM; ++i) {
        const u32x4 t = {alo[i][0], alo[i][1], ahi[i][0], ahi[i][1]};
        a[i] = __builtin_bit_cast(bf16x8, t);
      }
      // tap-major: the compiler's own waits for the ring fragments (issued after the hand-issued reads; LDS operations
      // return in order) cover the dY fragments
#pragma unroll
      for (int e = 0; e < 3; ++e) {
        const unsigned char* bt = ring + ((s + e - 1) & (NRING - 1)) * TILEB + fb;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
          const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(bt + c * 32));
          const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(bt + c * 32 + 16 * SPB));
          const bf16x8 b = tr_pair(lo, hi);
#pragma unroll
          for (int i = 0; i < WM; ++i) acc[i][e][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b, acc[i][e][c], 0, 0, 0);
        }
      }
    }
    stage_store(s + 2, Sold);
    // at most (this step's operations) outstanding => the dY tile of step s + 1 has landed (in-order completion)
    if (last_in) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(SIT + AIT) : "memory");
    else asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(SIT + AIT - 1) : "memory");
  };
  if (nsteps > 0) {
    if constexpr (PRO != 0) __syncthreads();                         // the prologue table
    stage_load(s_lo - 1, sg[0]);
    dma_a(s_lo, 0);
    dma_a(s_lo + 1, 1);
    stage_store(s_lo - 1, sg[0]);
    stage_load(s_lo, sg[0]);
    stage_store(s_lo, sg[0]);
    stage_load(s_lo + 1, sg[0]);
    stage_store(s_lo + 1, sg[0]);
    stage_load(s_lo + 2, sg[1]);                                     // stored at the end of the first step
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(SIT) : "memory");      // everything but that last request
    __syncthreads();
  }
  for (int k = 0; k < nsteps; k += 2) {
    const int ab = k % 3;
    step(s_lo + k, ab, sg[0], sg[1]);
    if (k + 1 < nsteps) step(s_lo + k + 1, ab == 2 ? 0 : ab + 1, sg[1], sg[0]);
  }
  // ---- partial tile: part[slice][mtiles * BM][3 * Cin_p]; C/D: col = lane & 15, rows (lane >> 4) * 4 + r
  const size_t ldp = (size_t)3 * g.Cin_p;
  float* pt = part + (size_t)kind * kind_stride + ((size_t)slice * g.mtiles * BM + m0 + wm * WM * 16) * ldp;
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    const int ch = c0 + wn * NC * 16 + c * 16;
    if (ch < g.Cin_p) {
#pragma unroll
      for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int e = 0; e < 3; ++e)
#pragma unroll
          for (int r = 0; r < 4; ++r) pt[(size_t)(i * 16 + fg * 4 + r) * ldp + (size_t)e * g.Cin_p + ch + fi] = acc[i][e][c][r];
    }
  }
}

// ---- host side ------------------------------------------------------------------------------------------------------
// (kt, kh, kw) = (3, 1, 1), stride 1, padding (1, 0, 0).  SELAVI_CL16_WGT=0 sends everything to the general kernel.
bool wgrad_t_plan(int N, int T, int H, int W, int Cin_p, int Cin, int Cout_p, int kt, int kh, int kw, int st, int sh, int sw,
                  int pt, int ph, int pw, int To, int Ho, int Wo, int* wm, int* nc, ClWgradT* out) {
  static int enabled = -1;
  if (enabled < 0) {
    const char* e = getenv("SELAVI_CL16_WGT");
    enabled = !(e && e[0] == '0');
  }
  if (!enabled) return false;
  if (kt != 3 || kh != 1 || kw != 1 || st != 1 || sh != 1 || sw != 1 || pt != 1 || ph != 0 || pw != 0) return false;
  if (To != T || Ho != H || Wo != W || T < 2) return false;
  ClWgradT g;
  g.N = N; g.T = T; g.HW = H * W; g.Cin_p = Cin_p; g.Cin = Cin; g.Cout_p = Cout_p;
  g.PB = (g.HW + 31) / 32;
  int w, c;
  if (Cout_p <= 64) {                                             // the whole N extent of the 64 x 160 layers in one block
    w = 2;
    c = Cin_p >= 160 ? 5 : (Cin_p >= 64 ? 2 : 1);
  } else {
    long long best_rows = 1LL << 60;
    w = 5;
    for (int t = 5; t >= 4; --t) {
      const long long rows = (long long)((Cout_p + 32 * t - 1) / (32 * t)) * 32 * t;
      if (rows < best_rows) {
        best_rows = rows;
        w = t;
      }
    }
    c = 2;
  }
  if (c == 1) return false;                                        // (no instantiation: narrow inputs stay on the general kernel)
  g.mtiles = (Cout_p + 32 * w - 1) / (32 * w);
  g.groups = (Cin_p + 32 * c - 1) / (32 * c);
  const long long steps = (long long)N * g.PB * (T + 1);
  if (steps >= (1LL << 30)) return false;
  const long long base = (long long)g.mtiles * g.groups;
  // one round of resident blocks: 2 per CU for (2,5) and (5,2) (199-203 / 177 registers), 3 for (4,2) (149 registers,
  // 48 KB LDS), 4 for (2,2) (97 registers, 36 KB)
  const long long slots = (w == 2 && c == 2) ? 1024 : (w == 4 ? 768 : 512);
  long long ksl = slots / base;
  if (ksl > steps / 64) ksl = steps / 64;                        // >= 64 steps per slice
  if (ksl < 1) ksl = 1;
  g.sper = (int)((steps + ksl - 1) / ksl);
  g.kslices = (int)((steps + g.sper - 1) / g.sper);
  *wm = w;
  *nc = c;
  *out = g;
  return true;
}

size_t wgrad_t_ws_bytes(const ClWgradT& g, int wm) { return (size_t)g.kslices * g.mtiles * wm * 32 * 3 * g.Cin_p * sizeof(float); }

template <int WM, int NC>
static void wgrad_t_launch_one(const ClWgradT& g, const void* dy, const void* x, const float* in_ss, float* part, hipStream_t st) {
  const FastDiv dT1 = make_fastdiv(g.T + 1), dPB = make_fastdiv(g.PB);
  const unsigned blocks = (unsigned)(g.kslices * g.mtiles * g.groups);
  if (in_ss)
    hipLaunchKernelGGL((cl16_wgrad_t_kernel<WM, NC, 1>), dim3(blocks), dim3(256), 0, st, (const unsigned short*)dy,
                       (const unsigned short*)x, in_ss, part, g, dT1, dPB, (size_t)0);
  else
    hipLaunchKernelGGL((cl16_wgrad_t_kernel<WM, NC, 0>), dim3(blocks), dim3(256), 0, st, (const unsigned short*)dy,
                       (const unsigned short*)x, in_ss, part, g, dT1, dPB, (size_t)0);
}

// PRO 2: both kinds of every (tile, slice), partials [2][slices][rows][3 Cin_p] (csrc/wgrad_cl16_t2.hip)
template <int WM, int NC>
static void wgrad_t_launch_dual_one(const ClWgradT& g, const void* dy, const void* x, const float* in_ss, float* part,
                                    size_t kind_stride, hipStream_t st) {
  const FastDiv dT1 = make_fastdiv(g.T + 1), dPB = make_fastdiv(g.PB);
  const unsigned blocks = (unsigned)(2 * g.kslices * g.mtiles * g.groups);
  hipLaunchKernelGGL((cl16_wgrad_t_kernel<WM, NC, 2>), dim3(blocks), dim3(256), 0, st, (const unsigned short*)dy,
                     (const unsigned short*)x, in_ss, part, g, dT1, dPB, kind_stride);
}

void wgrad_t_launch_dual(const ClWgradT& g, int wm, int nc, const void* dy, const void* x, const float* in_ss, float* part,
                         size_t kind_stride, hipStream_t st) {
  if (wm == 2 && nc == 5) wgrad_t_launch_dual_one<2, 5>(g, dy, x, in_ss, part, kind_stride, st);
  else if (wm == 2) wgrad_t_launch_dual_one<2, 2>(g, dy, x, in_ss, part, kind_stride, st);
  else if (wm == 4) wgrad_t_launch_dual_one<4, 2>(g, dy, x, in_ss, part, kind_stride, st);
  else wgrad_t_launch_dual_one<5, 2>(g, dy, x, in_ss, part, kind_stride, st);
}

void wgrad_t_launch(const ClWgradT& g, int wm, int nc, const void* dy, const void* x, const float* in_ss, float* part, hipStream_t st) {
  if (wm == 2 && nc == 5) wgrad_t_launch_one<2, 5>(g, dy, x, in_ss, part, st);
  else if (wm == 2) wgrad_t_launch_one<2, 2>(g, dy, x, in_ss, part, st);
  else if (wm == 4) wgrad_t_launch_one<4, 2>(g, dy, x, in_ss, part, st);
  else wgrad_t_launch_one<5, 2>(g, dy, x, in_ss, part, st);
}

}  // namespace slv
