// 16-bit MFMA path: the layer-1 spatial convolution Conv3d(64 -> 144, (1,3,3), stride 1, "same") -- the heaviest launch
// of the trunk (a fifth of the forward's bytes, bench.py's cfg5 roofline kernel) -- with the WEIGHTS RESIDENT IN
// REGISTERS and the data movement on a wave of its own.
//
// By arithmetic intensity (128 FLOP/B) this conv is bound by HBM; the tile kernel (csrc/conv_cl16_s3.hip) spends its
// time re-streaming the 166 KB weight tensor through LDS for every 128 positions (LDS-DMA issue + a barrier per tap),
// on per-tap border masks, and on prologue / epilogue phases that nothing overlaps.  Here:
//   * one workgroup of 4 waves per CU, persistent over 8 x 8-pixel tiles of one frame;
//   * waves 0-2 are MFMA waves: wave w owns output channels 48 w .. 48 w + 47 and holds their whole weight slice --
//     9 taps x 2 chunks x 3 tiles = 54 A fragments = 216 registers, in the accumulator half of the register file
//     (asm MFMAs with an "a" operand) -- for the kernel's lifetime.  Per tile: 72 ds_read_b128 of the input patch
//     (one base register, every tap / chunk / fragment an immediate offset), 216 MFMAs;
//     the tile's epilogue rides BETWEEN the MFMAs of the next tile (two accumulator sets): 12 items = round one
//     accumulator tile to bf16 into the workgroup's OUTPUT TILE in LDS (64 pixels x 160 channels, double buffered) and
//     add the rounded values to the per-lane running sums of the BatchNorm statistics;
//   * wave 3 is the data-movement wave, BOTH directions: it fetches the 10 x 10-pixel patch of the NEXT tile (16-byte
//     pieces, each lane a fixed channel piece, so BatchNorm + ReLU coefficients stay in registers; border handling from
//     four precomputed bit masks), applies BatchNorm + ReLU once per element and writes zeros for the halo outside the
//     image (no per-tap masks anywhere); and it moves the output tile finished two tiles ago to memory -- a tile row is 8
//     pixels x 320 bytes CONTIGUOUS, every store instruction writes 1 KB of whole cache lines (round 5; before, each MFMA
//     wave stored its own 96-byte segment of every pixel row -- partial lines from three waves at three times -- with the
//     LDS reads and the address arithmetic of 8 stores riding between its MFMAs: 0.58 -> 0.48 ms without those stores);
//   * patch and output tile are double buffered in LDS (one workgroup per CU: 92 KB of the 160), ONE barrier per tile.
// LDS patch: pixel (py, px) of the 10 x 10 patch at ((py * 16 + px) * 160) bytes -- pitch 16, rows of 128 + 32 bytes:
// found by enumeration to be conflict-free for ds_read_b128 fragments whose 16 positions are 2 tile rows x 8 columns,
// for every tap shift, without any XOR (so taps stay immediates).
#include "cl16.hpp"
#include "../../include/selavi_hip.h"

namespace slv {

constexpr int SR_T = 8;                               // tile edge (pixels)
constexpr int SR_PW = SR_T + 2;                       // patch edge
constexpr int SR_PITCH = 16, SR_ROWB = 160;           // LDS patch geometry (see above)
constexpr int SR_PATCH = SR_PW * SR_PITCH * SR_ROWB;  // 25 600 B
constexpr int SR_CIN = 64, SR_COUT = 144, SR_COUTP = 160;
constexpr int SR_OROW = SR_COUTP * 2 + 16;            // bytes per pixel row of the output tile (pitch 84 dwords: the 8-byte
                                                      // writes of a half wave -- 16 pixels x 2 k-groups -- hit 64 distinct banks)
constexpr int SR_OUT = SR_T * SR_T * SR_OROW;         // 21 504 B per output tile
constexpr int SR_LDS = 2 * SR_PATCH + 2 * SR_OUT;
constexpr int SR_NIT = (SR_PW * SR_PW + 7) / 8;       // 13 load instructions cover the patch (8 rows of 8 pieces each)
constexpr int SR_OPIECES = SR_T * SR_T * (SR_COUTP * 2 / 16);   // 1 280 16-byte pieces of an output tile
constexpr int SR_ONIT = SR_OPIECES / 64;              // = 20 store instructions of one wave

#ifndef SLV_SR_ABL
#define SLV_SR_ABL 0      // timing ablations (results wrong): 1 no MFMAs (a VALU xor keeps the operands alive), 2 no global
#endif                    // patch loads, 3 no output stores, 4 no epilogue at all, 5 no sched_barrier between MFMA groups, 6 two fragment sets per tile, 7 no statistics instructions
__device__ __forceinline__ void sr_mfma(f32x4& acc, const bf16x8& a, const bf16x8& b) {
#if SLV_SR_ABL == 1
  acc[0] += __builtin_bit_cast(float, __builtin_bit_cast(u32x4, b)[0]);
  asm volatile("" :: "a"(a));
#else
  asm("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc) : "a"(a), "v"(b));
#endif
}
__device__ __forceinline__ void sr_mfma0(f32x4& acc, const bf16x8& a, const bf16x8& b) {
#if SLV_SR_ABL == 1
  acc = (f32x4){__builtin_bit_cast(float, __builtin_bit_cast(u32x4, b)[0]), 0.f, 0.f, 0.f};
  asm volatile("" :: "a"(a));
#else
  asm("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=v"(acc) : "a"(a), "v"(b));
#endif
}
__device__ __forceinline__ void sr_add(float& a, float v) { asm volatile("v_add_f32 %0, %0, %1" : "+v"(a) : "v"(v)); }
__device__ __forceinline__ void sr_sq(float& q, float v) { asm volatile("v_fma_f32 %0, %1, %1, %0" : "+v"(q) : "v"(v)); }
__device__ __forceinline__ void sr_barrier() {        // LDS traffic of this wave complete, then the workgroup barrier;
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // global loads / stores stay in flight across it
  __builtin_amdgcn_s_barrier();
}

#ifdef SLV_SR_TRACE       // timing only (tools/sr_trace.py): per wave, s_memtime ticks summed over the tiles per section, written
#define SR_STAMP(i) do { const unsigned long long t_ = __builtin_readcyclecounter(); trc[i] += t_ - tlast; tlast = t_; } while (0)
#define SR_TRACE_DECL unsigned long long trc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = __builtin_readcyclecounter()
#define SR_TRACE_RESET tlast = __builtin_readcyclecounter()
#define SR_TRACE_OUT do { if (lane == 0) { unsigned long long* o_ = (unsigned long long*)y + (blockIdx.x * 4 + wave) * 8; for (int i_ = 0; i_ < 8; ++i_) o_[i_] = trc[i_]; } } while (0)
#else                     // over the head of y at the end of the kernel (the first output rows are garbage afterwards)
#define SR_STAMP(i) do { } while (0)
#define SR_TRACE_DECL do { } while (0)
#define SR_TRACE_RESET do { } while (0)
#define SR_TRACE_OUT do { } while (0)
#endif

struct SrTile {
  int y0, x0;
  unsigned fpos;                                      // position of the frame's pixel (0, 0)
};

// PRO 1: rows are read as relu(x * s + h).  EPI 1: per-channel sum / sum of squares of the rounded outputs, one partial
// per workgroup: stat_sum / stat_sq [Cout][gridDim.x].
template <int PRO, int EPI>
__global__ __launch_bounds__(256, 1) void conv_cl16_sr_kernel(const unsigned short* __restrict__ x,
                                                             const unsigned short* __restrict__ wl,
                                                             unsigned short* __restrict__ y,
                                                             const float* __restrict__ in_ss,
                                                             float* __restrict__ stat_sum, float* __restrict__ stat_sq,
                                                             ClConv g, int ntiles, int th, int tw) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  unsigned char* const patch = lds;                   // [2][SR_PATCH]
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int fr = lane & 15, fk = lane >> 4;
  const int nt = blockIdx.x < (unsigned)ntiles ? (ntiles - 1 - (int)blockIdx.x) / (int)gridDim.x + 1 : 0;
  const int H = g.Hi, W = g.Wi;
  // tile id -> (frame, tile row, tile column) by multiply-high with the reciprocals (exact: ids stay far below 2^32 / divisor).
  // As plain divisions the decode was ~70 instructions, once per tile in every MFMA wave that masks ragged tiles and twice per
  // tile in the data-movement wave: 300 of the 4 600 cycles of a tile with statistics.
  const unsigned per_u = (unsigned)(th * tw), mper = cl_recip(per_u), mtw = cl_recip((unsigned)tw);
  auto tile_of = [&](int k) __attribute__((always_inline)) {
    SrTile t;
    const unsigned idu = blockIdx.x + (unsigned)k * gridDim.x;
    const int per = (int)per_u, f = (int)cl_div(idu, mper), rem = (int)idu - f * per, ty = (int)cl_div((unsigned)rem, mtw);
    t.y0 = ty * SR_T;
    t.x0 = (rem - ty * tw) * SR_T;
    t.fpos = (unsigned)f * H * W;
    return t;
  };

  if (wave < 3) {
    // ======================================================================== MFMA waves
    bf16x8 A[9][2][3];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int i = 0; i < 3; ++i)
          A[t][c][i] = *(const bf16x8*)(wl + ((size_t)((t * 2 + c) * SR_COUT + (wave * 3 + i) * 16 + fr) * 32 + fk * 8));
    // the workgroup's output tiles [2][64 pixels][SR_OROW]: zeroed once (the 16 padding channels 144..159 are never
    // written again and reach memory as zeros), filled by the MFMA waves, moved to memory by the data-movement wave
    unsigned char* const outb = lds + 2 * SR_PATCH;
    for (int i = (wave * 64 + lane) * 16; i < 2 * SR_OUT; i += 3 * 64 * 16) *(u32x4*)(outb + i) = (u32x4){0u, 0u, 0u, 0u};
    // lane part of a fragment read: position fr of fragment nn is tile pixel (2 nn + (fr >> 3), fr & 7)
    const int lbase = (((fr >> 3) * SR_PITCH + (fr & 7)) * SR_ROWB) + fk * 16;
    const int obase = fr * SR_OROW + wave * 96 + fk * 8;
    // BatchNorm statistics: per-lane running sums of the ROUNDED outputs and their squares (channel (3 wave + i) * 16 +
    // 4 fk + r, over this lane's pixel column fr), reduced over the 16 lanes of a DPP row once, at the end of the kernel
    float stS[3][4], stQ[3][4];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) stS[i][r] = stQ[i][r] = 0.f;
    // The epilogue of tile n - 1 (round to bf16 into the output tile, statistics) is cut into single instructions and ONE
    // rides behind each MFMA of tile n (two accumulator sets, the loop unrolled by two).  Measured on this chip
    // (tools/proto/mfma_fill_bench.hip, one wave per SIMD): back-to-back v_mfma_f32_16x16x32_bf16 issue every 18.0 cycles;
    // one independent VALU instruction behind each costs +0.7 cycles, two +1.0, THREE +5, four +13 -- and a taken branch
    // ~30.  Rounds 3-4 placed a 25-instruction item behind every 7th MFMA: 1 000 of the tile's 5 100 cycles
    // (tools/sr_trace.py); three-instruction steps behind every third MFMA still cost 630.
    constexpr int SR_SUB = EPI == 1 ? 17 : 3, SR_STEPS = 12 * SR_SUB;       // 204 (36) of the tile's 216 MFMA slots
    unsigned dlo = 0, dhi = 0;
    float dv = 0.f;
    unsigned keep[4] = {~0u, ~0u, ~0u, ~0u};          // ragged tiles: pixels outside the image count as zero in the statistics
    auto drain_step = [&](int st, f32x4 (&pv)[3][4], unsigned char* ost) __attribute__((always_inline)) {
      const int a = st / SR_SUB, sub = st - a * SR_SUB, i = a >> 2, nn = a & 3;
      if (sub == 0) dlo = pack_bf2(pv[i][nn][0], pv[i][nn][1]);
      else if (sub == 1) dhi = pack_bf2(pv[i][nn][2], pv[i][nn][3]);
      else if constexpr (EPI == 0) {
        *(uint2*)(ost + obase + nn * 16 * SR_OROW + i * 32) = make_uint2(dlo, dhi);
      } else {                                        // (asm: left to itself hipcc SLP-packs adjacent sums into v_pk_add_f32 /
        if (sub == 2) dlo &= keep[nn];                // v_pk_fma_f32: an anti-lever beside MFMAs)
        else if (sub == 3) dhi &= keep[nn];
        else if (sub == 4) *(uint2*)(ost + obase + nn * 16 * SR_OROW + i * 32) = make_uint2(dlo, dhi);
        else {
          const int e = (sub - 5) / 3, w = (sub - 5) % 3;      // element e of the tile's four, instruction w of its three
          if (SLV_SR_ABL == 7) {                      // (ablation: the statistics' slots stay empty)
          } else if (w == 0) dv = e == 0 ? bf_lo(dlo) : e == 1 ? bf_hi(dlo) : e == 2 ? bf_lo(dhi) : bf_hi(dhi);
          else if (w == 1) sr_add(stS[i][e], dv);
          else sr_sq(stQ[i][e], dv);
        }
      }
    };
    auto set_keep = [&](const SrTile& tl) __attribute__((always_inline)) {
      if constexpr (EPI == 1) {
#pragma unroll
        for (int nn = 0; nn < 4; ++nn)                // (position fr of fragment nn is tile pixel (2 nn + (fr >> 3), fr & 7))
          keep[nn] = (tl.y0 + 2 * nn + (fr >> 3) < H && tl.x0 + (fr & 7) < W) ? ~0u : 0u;
      }
    };
    SR_TRACE_DECL;
    auto tile_step = [&](int n, f32x4 (&acc)[3][4], f32x4 (&pv)[3][4]) __attribute__((always_inline)) {
      const unsigned char* src = patch + (n & 1) * SR_PATCH + lbase;
      // (at n = 0 the other accumulator set is all zeros: zeros into an output tile nobody stores, zeros to the statistics)
      const SrTile tl = tile_of(n > 0 ? n - 1 : 0);
      unsigned char* const ost = outb + ((n & 1) ^ 1) * SR_OUT;   // tile n - 1's output tile
      set_keep(tl);
      // 18 MFMA groups (tap, chunk) of 12; the four fragments of group g + 1 are requested behind MFMAs 0, 2, 4, 6 of
      // group g (two register sets): no MFMA waits for an LDS read, no slot carries more than two instructions
      bf16x8 b[2][4];
      auto read_b1 = [&](int gi, int nn, bf16x8* to) __attribute__((always_inline)) {
        const int t = gi >> 1, c = gi & 1;
        to[nn] = *(const bf16x8*)(src + ((2 * nn + t / 3) * SR_PITCH + (t % 3)) * SR_ROWB + c * 64);
      };
#pragma unroll
      for (int nn = 0; nn < 4; ++nn) read_b1(0, nn, b[0]);
      if (SLV_SR_ABL == 6) {                          // (ablation: no fragment reads inside the tile)
#pragma unroll
        for (int nn = 0; nn < 4; ++nn) read_b1(1, nn, b[1]);
      }
#pragma unroll
      for (int gi = 0; gi < 18; ++gi) {
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
          for (int nn = 0; nn < 4; ++nn) {
            if (gi == 0) sr_mfma0(acc[i][nn], A[gi >> 1][gi & 1][i], b[gi & 1][nn]);
            else sr_mfma(acc[i][nn], A[gi >> 1][gi & 1][i], b[gi & 1][nn]);
            const int q = i * 4 + nn;                 // MFMA index within the group
            if (gi + 1 < 18 && q < 8 && (q & 1) == 0 && SLV_SR_ABL != 6) read_b1(gi + 1, q >> 1, b[(gi + 1) & 1]);
            // (tried: no epilogue instruction in the slots that also carry a fragment read and an s_waitcnt, two in the
            // slots 5 / 7 instead -- 4 635 -> 4 760 cycles per tile: the statistics cost ~4.5 cycles per VALU instruction
            // wherever they sit once the stream also carries the fragment reads)
            const int m = gi * 12 + q;
            if (m < SR_STEPS && SLV_SR_ABL != 4) drain_step(m, pv, ost);
            if (SLV_SR_ABL != 5) __builtin_amdgcn_sched_barrier(0);
          }
      }
      SR_STAMP(0);                                    // the tile's MFMAs + the previous tile's epilogue items
      sr_barrier();                                   // the patch buffer is free: the data-movement wave may refill it
      SR_STAMP(1);                                    // waiting for the other waves
    };
    f32x4 accA[3][4], accB[3][4];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int nn = 0; nn < 4; ++nn) accB[i][nn] = (f32x4){0.f, 0.f, 0.f, 0.f};
    sr_barrier();                                     // patch 0 is in LDS
    SR_TRACE_RESET;
    for (int n = 0; n < nt; n += 2) {
      tile_step(n, accA, accB);
      if (n + 1 < nt) tile_step(n + 1, accB, accA);
    }
    if (nt > 0) {                                     // the last tile's epilogue
      mfma_settle_nops();                             // 8-pass XDL result -> VALU read
#pragma unroll
      for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int nn = 0; nn < 4; ++nn) mfma_pin(accA[i][nn]), mfma_pin(accB[i][nn]);
      const SrTile tl = tile_of(nt - 1);
      unsigned char* const ost = outb + ((nt - 1) & 1) * SR_OUT;
      set_keep(tl);
      if ((nt - 1) & 1) {
#pragma unroll
        for (int it = 0; it < SR_STEPS; ++it) drain_step(it, accB, ost);
      } else {
#pragma unroll
        for (int it = 0; it < SR_STEPS; ++it) drain_step(it, accA, ost);
      }
    }
    sr_barrier();                                     // the last output tile is in LDS
    SR_STAMP(2);
    if constexpr (EPI == 1) {
#pragma unroll
      for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float a = row16_sum(stS[i][r]), q = row16_sum(stQ[i][r]);
          const int c = (wave * 3 + i) * 16 + fk * 4 + r;
          if (fr == 0) {
            stat_sum[(size_t)c * gridDim.x + blockIdx.x] = a;
            stat_sq[(size_t)c * gridDim.x + blockIdx.x] = q;
          }
        }
    }
    SR_TRACE_OUT;
    return;
  }

  // ========================================================================== data-movement wave
  const unsigned Ptot = (unsigned)g.N * g.Ti * H * W;
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, (int)(Ptot * (SR_CIN * 2u)), 0x00020000);
  // this lane's pieces of a patch: piece = lane & 7 of patch rows lr, lr + 8, ...
  const int piece = lane & 7, lr = lane >> 3;
  float ps[8], ph[8];
  if constexpr (PRO == 1) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int c = piece * 8 + e;
      ps[e] = c < g.Cin ? in_ss[c] : 0.f;
      ph[e] = c < g.Cin ? in_ss[g.Cin + c] : 0.f;
    }
  }
  // per piece (loop invariants): memory offset relative to the patch origin, LDS offset, and which borders it sits on
  unsigned goff[SR_NIT];
  int loff[SR_NIT];
  unsigned mtop = 0, mbot = 0, mleft = 0, mright = 0, mlive = 0;
#pragma unroll
  for (int i = 0; i < SR_NIT; ++i) {
    const int r = lr + 8 * i, py = r / SR_PW, px = r - py * SR_PW;
    const bool live = r < SR_PW * SR_PW;
    goff[i] = (unsigned)((py * W + px) * (SR_CIN * 2) + piece * 16);
    loff[i] = (live ? py * SR_PITCH + px : SR_PW * SR_PITCH - 1) * SR_ROWB + piece * 16;   // (rows past the patch: an unused pitch column)
    mlive |= (unsigned)live << i;
    mtop |= (unsigned)(py == 0) << i;
    mbot |= (unsigned)(py == SR_PW - 1) << i;
    mleft |= (unsigned)(px == 0) << i;
    mright |= (unsigned)(px == SR_PW - 1) << i;
  }
  // two register sets: patch k + 2 AND patch k + 3 are in flight while tile k is computed -- with one set the request
  // had less than one tile time (~2.5 us) to come back, and a late one put store_patch (BatchNorm + ReLU, 13 LDS writes)
  // on the workgroup's critical path (the prologue cost 0.09 ms of 0.6)
  u32x4 stA[SR_NIT], stB[SR_NIT];
  unsigned svA = 0, svB = 0;
  auto load_patch = [&](int k, u32x4 (&st)[SR_NIT], unsigned& stv) __attribute__((always_inline)) {
    const SrTile t = tile_of(k);
    // pieces inside the image.  Whole tiles (the common case): a piece is outside iff it lies on a patch border that
    // coincides with an image border -- four precomputed bit masks; ragged tiles and steps past the last tile: per piece
    const bool whole = k < nt && t.y0 + SR_T <= H && t.x0 + SR_T <= W;
    unsigned valid;
    if (whole) {
      valid = mlive & ~((t.y0 == 0 ? mtop : 0u) | (t.y0 + SR_T == H ? mbot : 0u) | (t.x0 == 0 ? mleft : 0u) |
                        (t.x0 + SR_T == W ? mright : 0u));
    } else {
      valid = 0;
      if (k < nt) {
#pragma unroll
        for (int i = 0; i < SR_NIT; ++i) {
          const int r = lr + 8 * i, py = r / SR_PW, px = r - py * SR_PW;
          const bool ok = (unsigned)(t.y0 - 1 + py) < (unsigned)H && (unsigned)(t.x0 - 1 + px) < (unsigned)W;
          valid |= (unsigned)ok << i;
        }
        valid &= mlive;
      }
    }
    stv = valid;
    const unsigned base = (t.fpos + (unsigned)((t.y0 - 1) * W + (t.x0 - 1))) * (SR_CIN * 2u);      // (wraps; used by valid pieces only)
#pragma unroll
    for (int i = 0; i < SR_NIT; ++i)
      st[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, (((valid >> i) & 1) && SLV_SR_ABL != 2) ? base + goff[i] : 0xFFFFFFF0u, 0, 0));
  };
  auto store_patch = [&](int buf, const u32x4 (&st)[SR_NIT], const unsigned& stv) __attribute__((always_inline)) {
    unsigned char* dst = patch + buf * SR_PATCH;
#pragma unroll
    for (int i = 0; i < SR_NIT; ++i) {
      u32x4 v = st[i];
      if constexpr (PRO == 1) {                       // zero padding AFTER the affine: the halo outside the image is zero
        const u32x4 a = affine_relu8(v, ps, ph);
        v = ((stv >> i) & 1) ? a : (u32x4){0u, 0u, 0u, 0u};
      }
      *(u32x4*)(dst + loff[i]) = v;
    }
  };
  // ---- the way out: output tile k (finished by the MFMA waves one barrier ago) -> memory.  Piece q = 64 it + lane of
  // the tile's 1 280: tile row q / 160 (8 pixels x 320 B, contiguous in memory), then pixel, then 16-byte piece
  const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc((void*)y, 0, (int)(Ptot * (SR_COUTP * 2u)), 0x00020000);
  const unsigned char* const outb = lds + 2 * SR_PATCH;
  unsigned oglo[SR_ONIT], opk[SR_ONIT];
#pragma unroll
  for (int it = 0; it < SR_ONIT; ++it) {
    const int q = it * 64 + lane, row = q / 160, rem = q - row * 160, px = rem / 20, pc = rem - px * 20;
    oglo[it] = (unsigned)((row * W + px) * (SR_COUTP * 2) + pc * 16);
    opk[it] = (unsigned)((row * SR_T + px) * SR_OROW + pc * 16) | (unsigned)row << 16 | (unsigned)px << 24;
  }
  auto store_out = [&](int k) __attribute__((always_inline)) {
    const SrTile t = tile_of(k > 0 ? k : 0);
    const unsigned char* src = outb + (k & 1) * SR_OUT;
    const unsigned base = (t.fpos + (unsigned)(t.y0 * W + t.x0)) * (SR_COUTP * 2u);
    const int hy = k >= 0 && SLV_SR_ABL != 3 ? H - t.y0 : 0, hx = W - t.x0;      // rows / columns of the tile inside the image
    const bool whole = hy >= SR_T && hx >= SR_T;      // (the common case: the tile's base as the scalar offset, no VALU per store)
#pragma unroll
    for (int h = 0; h < SR_ONIT; h += SR_ONIT / 2) {
      u32x4 v[SR_ONIT / 2];
#pragma unroll
      for (int it = 0; it < SR_ONIT / 2; ++it) v[it] = *(const u32x4*)(src + (opk[h + it] & 0xFFFFu));
      if (whole) {
#pragma unroll
        for (int it = 0; it < SR_ONIT / 2; ++it) __builtin_amdgcn_raw_buffer_store_b128(v[it], ry, oglo[h + it], base, CL_NT);
      } else {
#pragma unroll
        for (int it = 0; it < SR_ONIT / 2; ++it) {
          const bool ok = (int)((opk[h + it] >> 16) & 0xFFu) < hy && (int)(opk[h + it] >> 24) < hx;
          __builtin_amdgcn_raw_buffer_store_b128(v[it], ry, ok ? base + oglo[h + it] : 0xFFFFFFF0u, 0, CL_NT);
        }
      }
    }
  };
  // ---- pipeline head: patch 0 into buffer 0, patches 1 and 2 requested
  load_patch(0, stA, svA);
  load_patch(1, stB, svB);
  store_patch(0, stA, svA);
  load_patch(2, stA, svA);
  sr_barrier();
  SR_TRACE_DECL;
  auto phase = [&](int n, u32x4 (&st)[SR_NIT], unsigned& stv) __attribute__((always_inline)) {
#ifdef SLV_SR_TRACE
    asm volatile("s_waitcnt vmcnt(13)" ::: "memory");
    SR_STAMP(0);                                      // waiting for patch n + 1
#endif
    store_patch((n + 1) & 1, st, stv);                // patch n + 1 (requested two steps ago) -> the buffer tile n - 1 used
    SR_STAMP(1);
    store_out(n - 2);                                 // output tile n - 2: complete since the last barrier
    SR_STAMP(2);
    // (stores BEFORE the loads: vmcnt counts both, the compiler's counted waits for patch n + 2 in the next step assume only
    // the 13 loads of patch n + 3 are younger -- with the 20 stores behind them the wait would reach into patch n + 3)
    load_patch(n + 3, st, stv);                       // patch n + 3: in flight across two barriers
    SR_STAMP(3);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    SR_STAMP(4);
    sr_barrier();
    SR_STAMP(5);
  };
  for (int n = 0; n < nt; n += 2) {
    phase(n, stB, svB);
    if (n + 1 < nt) phase(n + 1, stA, svA);
  }
  store_out(nt - 2);
  sr_barrier();                                       // the MFMA waves have written the last output tile
  store_out(nt - 1);
  SR_STAMP(6);
  SR_TRACE_OUT;
}

static bool sr_enabled() {
  static const bool enabled = []() {
    const char* e = getenv("SELAVI_CL16_SR");
    return !(e && e[0] == '0');
  }();
  return enabled;
}

// Conv3d(64 -> 144, (1,3,3)), stride 1, padding (0,1,1), forward tap order (tap t = slab t at offset (t / 3 - 1, t % 3 - 1)).
bool cl16_sr_applies(const ClConv& g) {
  if (!sr_enabled() || g.ntaps != 9) return false;
  if (g.Cin_p != SR_CIN || g.Mrows != SR_COUT || g.Cout != SR_COUT || g.Cout_p != SR_COUTP) return false;
  if (g.Lt != g.Ti || g.Lh != g.Hi || g.Lw != g.Wi || g.To != g.Ti || g.Ho != g.Hi || g.Wo != g.Wi) return false;
  if (g.bmt != 1 || g.bmh != 1 || g.bmw != 1 || g.omt != 1 || g.omh != 1 || g.omw != 1 || g.oot || g.ooh || g.oow) return false;
  for (int t = 0; t < 9; ++t) {
    const int dt = (g.tap[t] & 15) - 8 + g.bot, dh = ((g.tap[t] >> 4) & 15) - 8 + g.boh, dw = ((g.tap[t] >> 8) & 15) - 8 + g.bow;
    if (dt != 0 || dh != t / 3 - 1 || dw != t % 3 - 1 || (g.tap[t] >> 12) != t) return false;
  }
  if ((long long)g.N * g.Ti * g.Hi * g.Wi * SR_COUTP * 2 >= 0xFFFFFFF0LL) return false;
  {                                                   // the tile decode by multiply-high is exact for ids < 2^32 / (tiles per frame)
    const long long per = (long long)((g.Hi + SR_T - 1) / SR_T) * ((g.Wi + SR_T - 1) / SR_T);
    if (((long long)g.N * g.Ti * per + 4096) * per >= 0xFFFFFFFFLL) return false;
  }
  return true;
}

static int sr_tiles(const ClConv& g) { return g.N * g.Ti * ((g.Hi + SR_T - 1) / SR_T) * ((g.Wi + SR_T - 1) / SR_T); }
static int sr_grid(const ClConv& g) {
  static const int blocks = []() {
    const char* e = getenv("SELAVI_CL16_SR_BLOCKS");
    return e ? atoi(e) : 256;                         // one persistent workgroup per CU
  }();
  const int t = sr_tiles(g);
  return t < blocks ? t : blocks;
}
int cl16_sr_slots(const ClConv& g) { return sr_grid(g); }

template <int PRO, int EPI>
static int sr_launch_one(const ClConv& g, const void* x, const void* wl, void* y, const float* in_ss, float* stat_sum,
                         float* stat_sq, hipStream_t st) {
  static bool attr_set = false;
  if (!attr_set) {
    SLV_HIP(hipFuncSetAttribute((const void*)conv_cl16_sr_kernel<PRO, EPI>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                160 * 1024));
    attr_set = true;
  }
  const int th = (g.Hi + SR_T - 1) / SR_T, tw = (g.Wi + SR_T - 1) / SR_T;
  hipLaunchKernelGGL((conv_cl16_sr_kernel<PRO, EPI>), dim3(sr_grid(g)), dim3(256), SR_LDS, st, (const unsigned short*)x,
                     (const unsigned short*)wl, (unsigned short*)y, in_ss, stat_sum, stat_sq, g, sr_tiles(g), th, tw);
  return 0;
}

// returns 1 when the launch was taken, 0 when it does not apply, < 0 on error
int cl16_sr_try(const ClConv& g, const void* x, const void* wl, void* y, const float* in_ss, const float* scale_shift,
                const void* res, int relu, float* stat_sum, float* stat_sq, const ClBnr& bnr, hipStream_t st) {
  if (!cl16_sr_applies(g)) return 0;
  if (bnr.part) return fail(-2, "%s: the register-resident conv has no fused BatchNorm-backward sums", "slv_cl16_conv");
  if (scale_shift || res || relu) return 0;           // eval-mode epilogues: the tile kernel (no statistics involved)
  int rc = 0;
  const int pro = in_ss ? 1 : 0, epi = stat_sum ? 1 : 0;
  if (pro == 0 && epi == 0) rc = sr_launch_one<0, 0>(g, x, wl, y, in_ss, stat_sum, stat_sq, st);
  else if (pro == 1 && epi == 0) rc = sr_launch_one<1, 0>(g, x, wl, y, in_ss, stat_sum, stat_sq, st);
  else if (pro == 0 && epi == 1) rc = sr_launch_one<0, 1>(g, x, wl, y, in_ss, stat_sum, stat_sq, st);
  else rc = sr_launch_one<1, 1>(g, x, wl, y, in_ss, stat_sum, stat_sq, st);
  if (rc) return rc;
  rc = launch_check("slv_cl16_conv");
  return rc ? rc : 1;
}

}  // namespace slv
