// A11: word-timestamp alignment on device (HF:models/whisper/generation_whisper.py:241-381, :43-61, :64-115).
//
//   rows [n_prompt, n_rows) of the alignment-head cross-attention probabilities, cropped to n_cols[b]
//   columns -> z-score over the token axis (population std) -> median filter width 7 over time
//   (reflect padding) -> mean over heads -> DTW on the negated matrix -> jump times.
//
// The reference runs the DTW as an O(N*M) pure-Python double loop on the host.  Here the three
// dependencies of a cell lie on the two previous anti-diagonals, so one workgroup per stream sweeps
// the anti-diagonals with the rolling cost rows in LDS (one barrier per diagonal) and writes the
// 2-bit decisions to a trace plane; a single lane then walks the trace back.  Cost cells follow the
// reference's arithmetic exactly: float64 add of the (negated) matrix entry and the float32 running
// cost, rounded to float32 on store; strict '<' tie-breaking (diagonal, then up, else left).
#include "tw_common.h"

namespace {

__global__ void align_zscore_kernel(DtwArgs a) {
  const int b = blockIdx.z, hd = blockIdx.y;
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  const int M = a.n_cols[b];
  const int N = a.n_rows - a.n_prompt;
  if (j >= M) return;
  const float* src = a.align + (((long long)b * a.Ha + hd) * a.P + a.n_prompt) * a.T + j;
  float s = 0.f;
  for (int i = 0; i < N; ++i) s += src[(long long)i * a.T];
  const float mean = s / (float)N;
  float q = 0.f;
  for (int i = 0; i < N; ++i) { const float c = src[(long long)i * a.T] - mean; q += c * c; }
  const float sd = sqrtf(q / (float)N);
  float* dst = a.zbuf + (((long long)b * a.Ha + hd) * N) * a.T + j;
  for (int i = 0; i < N; ++i) dst[(long long)i * a.T] = (src[(long long)i * a.T] - mean) / sd;
}

__device__ __forceinline__ void cswap(float& x, float& y) {
  const float lo = fminf(x, y), hi = fmaxf(x, y);
  x = lo; y = hi;
}

__global__ void align_median_mean_kernel(DtwArgs a) {
  const int b = blockIdx.z, i = blockIdx.y;
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  const int M = a.n_cols[b];
  const int N = a.n_rows - a.n_prompt;
  if (j >= M) return;
  const int pad = a.median_width / 2;  // width 7 supported (config.median_filter_width default)
  float acc = 0.f;
  for (int hd = 0; hd < a.Ha; ++hd) {
    const float* row = a.zbuf + (((long long)b * a.Ha + hd) * N + i) * a.T;
    float v;
    if (M <= pad) {
      v = row[j];  // HF returns the input unfiltered when the axis is <= pad_width
    } else {
      float w[7];
#pragma unroll
      for (int k = 0; k < 7; ++k) {
        int jj = j + k - 3;
        if (jj < 0) jj = -jj;
        if (jj >= M) jj = 2 * (M - 1) - jj;
        w[k] = row[jj];
      }
      // sorting network for 7 elements (16 compare-exchanges); median = w[3]
      cswap(w[0], w[6]); cswap(w[2], w[3]); cswap(w[4], w[5]);
      cswap(w[0], w[2]); cswap(w[1], w[4]); cswap(w[3], w[6]);
      cswap(w[0], w[1]); cswap(w[2], w[5]); cswap(w[3], w[4]);
      cswap(w[1], w[2]); cswap(w[4], w[6]);
      cswap(w[2], w[3]); cswap(w[4], w[5]);
      cswap(w[1], w[2]); cswap(w[3], w[4]); cswap(w[5], w[6]);
      v = w[3];
    }
    acc += v;
  }
  a.mat[((long long)b * N + i) * a.T + j] = acc / (float)a.Ha;
}

__global__ __launch_bounds__(512) void dtw_kernel(DtwArgs a) {
  __shared__ float diag[3][512];
  __shared__ float jump[512];
  const int b = blockIdx.x;
  const int i = threadIdx.x;
  const int M = a.n_cols[b];
  const int N = a.n_rows - a.n_prompt;
  float* ts = a.out_ts + (long long)b * (a.n_rows + 1);
  if (N <= 0 || M <= 0) {
    // M == 0 (every column cropped away): the reference still runs its DTW, on an [N, 0] matrix - the back-trace walks the
    // token axis at time index -1 (HF:models/whisper/generation_whisper.py:88-115), so every generated token (and the
    // duplicate of the last one) gets -1 * time_precision; the prompt positions stay 0
    const float v = (N > 0) ? (float)(-1.0 * a.time_precision) : 0.f;
    for (int k = i; k < a.n_rows + 1; k += 512) ts[k] = k >= a.n_prompt ? v : 0.f;
    return;
  }
  const float* mat = a.mat + (long long)b * N * a.T;
  signed char* trace = a.trace + (long long)b * (long long)(a.P + 1) * (a.T + 1);
  const int W = M + 1;
  const float INF = INFINITY;
  for (int k = 0; k <= N + M; ++k) {
    float* cur = diag[k % 3];
    const float* p1 = diag[(k + 2) % 3];  // diagonal k-1
    const float* p2 = diag[(k + 1) % 3];  // diagonal k-2
    const int j = k - i;
    if (i <= N && j >= 0 && j <= M) {
      float c;
      if (i == 0 || j == 0) {
        c = (i == 0 && j == 0) ? 0.f : INF;
      } else {
        const float c0 = p2[i - 1], c1 = p1[i - 1], c2 = p1[i];
        float cm;
        signed char t;
        if (c0 < c1 && c0 < c2) { cm = c0; t = 0; }
        else if (c1 < c0 && c1 < c2) { cm = c1; t = 1; }
        else { cm = c2; t = 2; }
        c = (float)(-(double)mat[(long long)(i - 1) * a.T + (j - 1)] + (double)cm);
        trace[(long long)i * W + j] = t;
      }
      cur[i] = c;
    }
    __syncthreads();
  }
  // every thread's trace stores must be visible to the walker: same workgroup, barrier above + fence
  __threadfence_block();
  __syncthreads();
  if (i == 0) {
    int ii = N, jj = M;
    while (ii > 0 || jj > 0) {
      if (ii > 0) jump[ii - 1] = (float)((double)(jj - 1) * a.time_precision);
      int t;
      if (ii == 0) t = 2;
      else if (jj == 0) t = 1;
      else t = trace[(long long)ii * W + jj];
      if (t == 0) { --ii; --jj; }
      else if (t == 1) { --ii; }
      else { --jj; }
    }
  }
  __syncthreads();
  for (int k = i; k < a.n_rows + 1; k += 512) {
    float v = 0.f;
    if (k >= a.n_prompt) {
      const int r = k - a.n_prompt;
      v = jump[r < N ? r : N - 1];
    }
    ts[k] = v;
  }
}

}  // namespace

hipError_t launch_token_timestamps(const DtwArgs& a, hipStream_t st) {
  const int N = a.n_rows - a.n_prompt;
  if (a.B <= 0) return hipSuccess;
  if (N > 511 || a.median_width != 7) return hipErrorInvalidValue;
  if (N > 0) {
    dim3 g1((a.T + 255) / 256, a.Ha, a.B);
    hipLaunchKernelGGL(align_zscore_kernel, g1, dim3(256), 0, st, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    dim3 g2((a.T + 255) / 256, N, a.B);
    hipLaunchKernelGGL(align_median_mean_kernel, g2, dim3(256), 0, st, a);
    e = hipGetLastError();
    if (e != hipSuccess) return e;
  }
  hipLaunchKernelGGL(dtw_kernel, dim3(a.B), dim3(512), 0, st, a);
  return hipGetLastError();
}
