// A1: log-mel front end (HF:models/whisper/feature_extraction_whisper.py:135-168) on gfx950.
//
//   frame t = hann(400) * padded_pcm[t*160 - 200 .. t*160 + 200)   (centre=True, reflect padding of the
//   zero-padded n_samples-long signal; the last STFT frame is dropped)
//   power = |DFT_400(frame)|^2 (201 bins) -> mel = bank^T . power -> log10(max(mel,1e-10))
//   -> per-clip max -> max(x, gmax-8) -> (x+4)/4
//
// The 400-point DFT is evaluated directly in float64 (CDNA4 has full-rate-ish f64 FMA; the whole
// front end is < 1 % of a chunk's time), using the real-signal fold
//   Re X[k] = sum_{n=0..200} s[n] cos(2 pi k n / 400),   s[n] = w[n] + w[400-n]  (s[0]=w[0], s[200]=w[200])
//   Im X[k] = sum_{n=1..199} d[n] sin(2 pi k n / 400),   d[n] = w[n] - w[400-n]
// which halves the work.  One workgroup = 16 consecutive frames of one clip: the folded frames sit in
// LDS as [n][frame] so that a thread (= one frequency bin) reads them as wave-uniform broadcasts and
// reuses one twiddle for 16 frames; the sparse mel bank is applied from per-filter [lo,hi) ranges.
// Results are closer to exact arithmetic than the reference's float32 FFT (tests state the tolerance).
#include "tw_common.h"

namespace {

constexpr int NFFT = 400;
constexpr int HOPSZ = 160;
constexpr int NBIN = 201;
constexpr int FR = 16;  // frames per workgroup

__device__ __forceinline__ unsigned f2ord(float f) {
  const unsigned u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float ord2f(unsigned u) {
  return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u);
}

__global__ __launch_bounds__(256) void logmel_frames_kernel(const float* __restrict__ pcm, long long pcm_stride,
                                                             const int* __restrict__ n_valid, int n_samples,
                                                             int n_frames, int n_mels, LogmelTables tb,
                                                             float* __restrict__ logspec, unsigned* __restrict__ gmax) {
  __shared__ double xs[NBIN][FR];   // sum fold, later reused for power
  __shared__ double xd[NBIN][FR];   // difference fold
  __shared__ double tw[NFFT];
  __shared__ float red[4];
  const int b = blockIdx.y;
  const int t0 = blockIdx.x * FR;
  const int tid = threadIdx.x;
  const float* x = pcm + (long long)b * pcm_stride;
  const int nv = n_valid ? min(n_valid[b], n_samples) : n_samples;

  for (int i = tid; i < NFFT; i += 256) tw[i] = tb.twiddle[i];

  auto sample = [&](int t, int n) -> double {  // windowed sample n of frame t
    int idx = t * HOPSZ - NFFT / 2 + n;
    if (idx < 0) idx = -idx;
    if (idx >= n_samples) idx = 2 * (n_samples - 1) - idx;
    const float v = (idx < nv) ? x[idx] : 0.0f;
    return (double)v * tb.window[n];
  };
  for (int i = tid; i < NBIN * FR; i += 256) {
    const int n = i / FR, f = i % FR;
    const int t = t0 + f;
    double s = 0.0, d = 0.0;
    if (t < n_frames) {
      const double a = sample(t, n);
      if (n == 0 || n == NFFT / 2) {
        s = a;
      } else {
        const double c = sample(t, NFFT - n);
        s = a + c;
        d = a - c;
      }
    }
    xs[n][f] = s;
    xd[n][f] = d;
  }
  __syncthreads();

  double re[FR], im[FR];
#pragma unroll
  for (int f = 0; f < FR; ++f) { re[f] = 0.0; im[f] = 0.0; }
  const int k = tid;
  if (k < NBIN) {
    int idx = 0;  // (k*n) mod 400
    for (int n = 0; n < NBIN; ++n) {
      const double c = tw[idx];
      int is = idx + 300;  // sin(theta) = cos(theta - pi/2) = cos(2 pi (j-100)/400) = cos(2 pi (j+300)/400)
      if (is >= NFFT) is -= NFFT;
      const double s = tw[is];
#pragma unroll
      for (int f = 0; f < FR; ++f) {
        re[f] = fma(c, xs[n][f], re[f]);
        im[f] = fma(s, xd[n][f], im[f]);
      }
      idx += k;
      if (idx >= NFFT) idx -= NFFT;
    }
  }
  __syncthreads();
  if (k < NBIN) {
#pragma unroll
    for (int f = 0; f < FR; ++f) xs[k][f] = re[f] * re[f] + im[f] * im[f];
  }
  __syncthreads();

  float lmax = -3.0e38f;
  for (int i = tid; i < n_mels * FR; i += 256) {
    const int m = i / FR, f = i % FR;
    const int t = t0 + f;
    if (t >= n_frames) continue;
    double acc = 0.0;
    const int lo = tb.lo[m], hi = tb.hi[m];
    for (int kk = lo; kk < hi; ++kk) acc = fma((double)tb.bank[kk * n_mels + m], xs[kk][f], acc);
    const float lv = (float)log10(fmax(acc, 1e-10));
    logspec[((long long)b * n_mels + m) * n_frames + t] = lv;
    lmax = fmaxf(lmax, lv);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) lmax = fmaxf(lmax, __shfl_xor(lmax, o, 64));
  if ((tid & 63) == 0) red[tid >> 6] = lmax;
  __syncthreads();
  if (tid == 0) {
    const float m4 = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    atomicMax(gmax + b, f2ord(m4));
  }
}

template <typename T>
__global__ void logmel_finalize_kernel(const float* __restrict__ logspec, const unsigned* __restrict__ gmax,
                                       T* __restrict__ out, long long per_clip) {
  const int b = blockIdx.y;
  const float floor_v = ord2f(gmax[b]) - 8.0f;
  const long long base = (long long)b * per_clip;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < per_clip;
       i += (long long)gridDim.x * blockDim.x) {
    const float v = fmaxf(logspec[base + i], floor_v);
    out[base + i] = (T)((v + 4.0f) / 4.0f);
  }
}

}  // namespace

hipError_t launch_logmel(const float* pcm, long long pcm_stride, const int* n_valid_dev, int B, int n_samples,
                         int n_mels, const LogmelTables& tb, float* logspec_ws, unsigned* max_ws, void* out,
                         int out_dtype, hipStream_t st) {
  const int n_frames = n_samples / HOPSZ;
  if (B <= 0 || n_frames <= 0 || n_samples < NFFT) return hipErrorInvalidValue;
  hipError_t e = hipMemsetAsync(max_ws, 0, sizeof(unsigned) * B, st);
  if (e != hipSuccess) return e;
  dim3 grid((n_frames + FR - 1) / FR, B);
  hipLaunchKernelGGL(logmel_frames_kernel, grid, dim3(256), 0, st, pcm, pcm_stride, n_valid_dev, n_samples, n_frames,
                     n_mels, tb, logspec_ws, max_ws);
  e = hipGetLastError();
  if (e != hipSuccess) return e;
  const long long per_clip = (long long)n_mels * n_frames;
  dim3 g2((unsigned)min((per_clip + 255) / 256, (long long)1024), B);
  if (out_dtype == 1)
    hipLaunchKernelGGL(logmel_finalize_kernel<bf16_t>, g2, dim3(256), 0, st, logspec_ws, max_ws, (bf16_t*)out, per_clip);
  else if (out_dtype == 2)
    hipLaunchKernelGGL(logmel_finalize_kernel<f16_t>, g2, dim3(256), 0, st, logspec_ws, max_ws, (f16_t*)out, per_clip);
  else if (out_dtype == 0)
    hipLaunchKernelGGL(logmel_finalize_kernel<float>, g2, dim3(256), 0, st, logspec_ws, max_ws, (float*)out, per_clip);
  else
    return hipErrorInvalidValue;
  return hipGetLastError();
}
