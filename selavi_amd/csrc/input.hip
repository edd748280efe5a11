// Input pipeline on the device: what the reference's DataLoader workers do on the CPU for
// every clip, as two HBM-bound kernels over a whole batch.
//
//   slv_clip_augment   datasets/video_transforms.py:462-510  uint8 THWC frames -> normalised, short-side-resized
//                      (bilinear, align_corners=False), cropped, optionally flipped float32 CTHW clip.  One read
//                      of the source bytes, one write of the clip; the reference materialises four intermediates.
//   slv_logfbank       datasets/audio_utils.py:46-72 -> python_speech_features.logfbank (0.6): pre-emphasis, framing
//                      (rectangular window), |rfft|^2/nfft, triangular mel filterbank, log -- in float64 like numpy,
//                      stored as float32 [B][1][nfilt][frames].
#include "common.hpp"

namespace slv {

// ---- video --------------------------------------------------------------------------------------------------------
struct ClipDesc {           // one per clip, int64 x 8 on the device
  long long src_off;        // byte offset of this clip's T*H*W*3 frames in the source buffer
  long long H, W;           // source size
  long long nh, nw;         // size after the short-side resize (== H, W: no resize)
  long long y_off, x_off;   // crop origin in the resized image
  long long flip;
};

// torch's bilinear coefficients (aten UpSample.h, align_corners=False), in float32 with the source index formed by
// one fused multiply-add -- bit-identical to the CPU build the reference runs on (oracle/input_ref.py:_axis)
__device__ __forceinline__ void axis_coef(int dst, int n_in, int n_out, int& i0, int& i1, float& w0, float& w1) {
  const float scale = (float)n_in / (float)n_out;
  float src = __fmaf_rn(scale, (float)dst + 0.5f, -0.5f);
  src = fmaxf(src, 0.f);
  i0 = min((int)src, n_in - 1);
  w1 = fminf(fmaxf(src - (float)i0, 0.f), 1.f);
  w0 = 1.f - w1;
  i1 = i0 + (i0 < n_in - 1 ? 1 : 0);
}

__device__ __forceinline__ float norm_px(unsigned char v, float mean, float stdv) {
  return ((float)v / 255.0f - mean) / stdv;      // video_transforms.py:475-478, one rounding per step like torch
}

// grid (ceil(S*S/256), T, B); thread = one output pixel, all three channels (the source is channel-interleaved)
__global__ __launch_bounds__(256) void clip_augment_kernel(const unsigned char* __restrict__ src,
                                                           const ClipDesc* __restrict__ desc, float* __restrict__ out,
                                                           int T, int S, float m0, float m1, float m2, float s0,
                                                           float s1, float s2) {
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= S * S) return;
  const int oy = p / S, ox = p - oy * S, t = blockIdx.y, b = blockIdx.z;
  const ClipDesc d = desc[b];
  const int H = (int)d.H, W = (int)d.W, nh = (int)d.nh, nw = (int)d.nw;
  const int ry = oy + (int)d.y_off, rx = (d.flip ? S - 1 - ox : ox) + (int)d.x_off;
  const unsigned char* f = src + d.src_off + (size_t)t * H * W * 3;
  const float mean[3] = {m0, m1, m2}, stdv[3] = {s0, s1, s2};
  float* o = out + (((size_t)b * 3) * T + t) * S * S + p;
  const size_t cstride = (size_t)T * S * S;
  if (nh == H && nw == W) {                      // random_short_side_scale_jitter returned the images unchanged
    const unsigned char* q = f + ((size_t)ry * W + rx) * 3;
#pragma unroll
    for (int c = 0; c < 3; ++c) o[c * cstride] = norm_px(q[c], mean[c], stdv[c]);
    return;
  }
  int y0, y1, x0, x1;
  float wy0, wy1, wx0, wx1;
  axis_coef(ry, H, nh, y0, y1, wy0, wy1);
  axis_coef(rx, W, nw, x0, x1, wx0, wx1);
  const unsigned char *q00 = f + ((size_t)y0 * W + x0) * 3, *q01 = f + ((size_t)y0 * W + x1) * 3,
                      *q10 = f + ((size_t)y1 * W + x0) * 3, *q11 = f + ((size_t)y1 * W + x1) * 3;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float p00 = norm_px(q00[c], mean[c], stdv[c]), p01 = norm_px(q01[c], mean[c], stdv[c]);
    const float p10 = norm_px(q10[c], mean[c], stdv[c]), p11 = norm_px(q11[c], mean[c], stdv[c]);
    const float top = __fmaf_rn(p00, wx0, p01 * wx1), bot = __fmaf_rn(p10, wx0, p11 * wx1);
    o[c * cstride] = __fmaf_rn(top, wy0, bot * wy1);
  }
}

// ---- audio --------------------------------------------------------------------------------------------------------
constexpr int FB_MAX_FRAME = 2048, FB_MAX_NFFT = 2048;

// grid (nframes, B), 256 threads.  LDS: the pre-emphasised frame, the twiddle table, the power spectrum.
__global__ __launch_bounds__(256) void logfbank_kernel(const short* __restrict__ wav, const long long* __restrict__ start,
                                                       const double* __restrict__ volume, long long wav_stride,
                                                       int slen, int frame_len, int frame_step, int nfft, int nfilt,
                                                       const double* __restrict__ twiddle,   // cos[nfft], sin[nfft]
                                                       const int* __restrict__ bins,         // nfilt + 2
                                                       double preemph, int z_normalize, float* __restrict__ out,
                                                       int nframes) {
  extern __shared__ double lds[];
  double* x = lds;                         // frame_len
  double* tc = x + frame_len;              // nfft
  double* ts = tc + nfft;                  // nfft
  double* ps = ts + nfft;                  // nfft/2 + 1
  const int fr = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
  const short* w = wav + (size_t)b * wav_stride + start[b];
  const double vol = volume ? volume[b] : 1.0;
  const bool scaled = volume != nullptr;
  for (int i = tid; i < frame_len; i += 256) {
    const int n = fr * frame_step + i;     // sample index inside the (zero-padded) signal
    double v = 0.0;
    if (n < slen) {                        // sigproc.preemphasis: s[0], s[n] - coeff*s[n-1]
      const double cur = scaled ? (double)w[n] * vol : (double)w[n];
      if (n == 0) v = cur;
      else {
        const double prev = scaled ? (double)w[n - 1] * vol : (double)w[n - 1];
        v = cur - preemph * prev;
      }
    }
    x[i] = v;
  }
  for (int i = tid; i < nfft; i += 256) {
    tc[i] = twiddle[i];
    ts[i] = twiddle[nfft + i];
  }
  __syncthreads();
  const int nbin = nfft / 2 + 1, mask = nfft - 1;
  for (int k = tid; k < nbin; k += 256) {
    double re = 0.0, im = 0.0;
    int idx = 0;
    for (int n = 0; n < frame_len; ++n) {
      re = fma(x[n], tc[idx], re);
      im = fma(x[n], ts[idx], im);
      idx = (idx + k) & mask;
    }
    ps[k] = (re * re + im * im) * (1.0 / nfft);
  }
  __syncthreads();
  for (int j = tid; j < nfilt; j += 256) {
    const int b0 = bins[j], b1 = bins[j + 1], b2 = bins[j + 2];
    double acc = 0.0;
    for (int i = b0; i < b1; ++i) acc += ps[i] * ((double)(i - b0) / (double)(b1 - b0));
    for (int i = b1; i < b2; ++i) acc += ps[i] * ((double)(b2 - i) / (double)(b2 - b1));
    if (acc == 0.0) acc = 2.220446049250313e-16;   // numpy.finfo(float).eps
    float v = (float)log(acc);
    if (z_normalize) v = (v - 1.93f) / 17.89f;     // audio_utils.py:71-72
    out[((size_t)b * nfilt + j) * nframes + fr] = v;
  }
}

}  // namespace slv

extern "C" {

int slv_clip_augment(const void* frames_u8, const int64_t* desc, float* out, int B, int T, int S,
                     const float* mean3, const float* std3, void* stream) {
  using namespace slv;
  SLV_CHECK_ARG(frames_u8 && desc && out && mean3 && std3, "null pointer");
  SLV_CHECK_ARG(B > 0 && T > 0 && S > 0 && B <= 65535 && T <= 65535, "bad sizes");
  dim3 grid(cdiv((long)S * S, 256), T, B);
  hipLaunchKernelGGL(clip_augment_kernel, grid, dim3(256), 0, (hipStream_t)stream, (const unsigned char*)frames_u8,
                     (const ClipDesc*)desc, (float*)out, T, S, mean3[0], mean3[1], mean3[2], std3[0], std3[1],
                     std3[2]);
  SLV_LAUNCH_CHECK();
  return 0;
}

int32_t slv_logfbank_frames(int slen, int frame_len, int frame_step) {
  if (slen <= 0 || frame_len <= 0 || frame_step <= 0) return -1;
  if (slen <= frame_len) return 1;
  return 1 + (int)((slen - frame_len + frame_step - 1) / frame_step);
}

int slv_logfbank(const void* wav_i16, const int64_t* start_i64, const double* volume_f64, int64_t wav_stride, int B,
                 int slen, int frame_len, int frame_step, int nfft, int nfilt, const double* twiddle_f64,
                 const int32_t* bins_i32, double preemph, int z_normalize, float* out_f32, void* stream) {
  using namespace slv;
  SLV_CHECK_ARG(wav_i16 && start_i64 && twiddle_f64 && bins_i32 && out_f32, "null pointer");
  SLV_CHECK_ARG(B > 0 && B <= 65535 && slen > 0 && nfilt > 0, "bad sizes");
  SLV_CHECK_ARG(nfft > 0 && (nfft & (nfft - 1)) == 0 && nfft <= FB_MAX_NFFT, "nfft must be a power of two <= 2048");
  SLV_CHECK_ARG(frame_len > 0 && frame_len <= nfft && frame_len <= FB_MAX_FRAME && frame_step > 0,
                "frame length must not exceed nfft");
  const int nframes = slv_logfbank_frames(slen, frame_len, frame_step);
  const size_t lds = sizeof(double) * ((size_t)frame_len + 2 * (size_t)nfft + nfft / 2 + 1);
  hipLaunchKernelGGL(logfbank_kernel, dim3(nframes, B), dim3(256), lds, (hipStream_t)stream, (const short*)wav_i16,
                     (const long long*)start_i64, (const double*)volume_f64, wav_stride, slen, frame_len, frame_step,
                     nfft, nfilt, (const double*)twiddle_f64, (const int*)bins_i32, preemph, z_normalize,
                     (float*)out_f32, nframes);
  SLV_LAUNCH_CHECK();
  return 0;
}

}  // extern "C"
