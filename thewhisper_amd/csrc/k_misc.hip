// Small HBM-bound helper kernels: LayerNorm (wave-per-row, shuffle reductions), layout transposes feeding
// the conv-stem GEMMs, dtype conversion, weight repacking at load time, token+position embedding.
#include "tw_common.h"

namespace {

__device__ __forceinline__ float wave_sum(float v) { return tw_wave_sum(v); }

template <typename T> __device__ __forceinline__ float ldf(const T* p) { return (float)*p; }

template <typename T> __device__ __forceinline__ void load16(const T* p, float* out);
template <> __device__ __forceinline__ void load16<float>(const float* p, float* out) {
  const f32x4_t v = *reinterpret_cast<const f32x4_t*>(p);
  out[0] = v[0]; out[1] = v[1]; out[2] = v[2]; out[3] = v[3];
}
template <> __device__ __forceinline__ void load16<bf16_t>(const bf16_t* p, float* out) {
  const bf16x8_t v = *reinterpret_cast<const bf16x8_t*>(p);
#pragma unroll
  for (int i = 0; i < 8; ++i) out[i] = (float)v[i];
}
template <> __device__ __forceinline__ void load16<f16_t>(const f16_t* p, float* out) {
  const f16x8_t v = *reinterpret_cast<const f16x8_t*>(p);
#pragma unroll
  for (int i = 0; i < 8; ++i) out[i] = (float)v[i];
}
template <typename T> __device__ __forceinline__ void store16(T* p, const float* in);
template <> __device__ __forceinline__ void store16<float>(float* p, const float* in) {
  *reinterpret_cast<f32x4_t*>(p) = f32x4_t{in[0], in[1], in[2], in[3]};
}
template <> __device__ __forceinline__ void store16<bf16_t>(bf16_t* p, const float* in) {
  bf16x8_t v;
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = (bf16_t)in[i];
  *reinterpret_cast<bf16x8_t*>(p) = v;
}
template <> __device__ __forceinline__ void store16<f16_t>(f16_t* p, const float* in) {
  u32x4_t v;
#pragma unroll
  for (int i = 0; i < 4; ++i) v[i] = tw_pack2<f16_t>(in[2 * i], in[2 * i + 1]);   // saturating
  *reinterpret_cast<u32x4_t*>(p) = v;
}

// One wave per row; the row lives in registers between the mean and variance passes (two-pass
// LayerNorm like torch's, eps 1e-5, biased variance).
template <typename T>
__global__ __launch_bounds__(256) void layernorm_kernel(const T* __restrict__ x, const T* __restrict__ g,
                                                         const T* __restrict__ b, T* __restrict__ y, int rows, int d) {
  constexpr int E = ElemTraits<T>::kPer16B;
  constexpr int MAXV = 5;
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int nv = d / E;
  const T* xr = x + (long long)row * d;
  float v[MAXV][E];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int vi = lane + i * 64;
    if (vi < nv) {
      load16<T>(xr + vi * E, v[i]);
#pragma unroll
      for (int e = 0; e < E; ++e) s += v[i][e];
    }
  }
  const float mean = wave_sum(s) / (float)d;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int vi = lane + i * 64;
    if (vi < nv) {
#pragma unroll
      for (int e = 0; e < E; ++e) { const float c = v[i][e] - mean; q += c * c; }
    }
  }
  const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)d + 1e-5f);
  T* yr = y + (long long)row * d;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int vi = lane + i * 64;
    if (vi < nv) {
      float gg[E], bb[E], o[E];
      load16<T>(g + vi * E, gg);
      load16<T>(b + vi * E, bb);
#pragma unroll
      for (int e = 0; e < E; ++e) o[e] = (v[i][e] - mean) * rstd * gg[e] + bb[e];
      store16<T>(yr + vi * E, o);
    }
  }
}

template <typename TD, typename TS>
__global__ void mel_transpose_kernel(const TS* __restrict__ mel, TD* __restrict__ melT, int B, int n_mels, int F, int C) {
  // tile transpose through LDS: block = 32 frames x 32 channels
  __shared__ float tile[32][33];
  const int b = blockIdx.z;
  const int f0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 256 threads: ty 0..7
  for (int r = ty; r < 32; r += 8) {
    const int c = c0 + r, f = f0 + tx;
    float v = 0.f;
    if (c < n_mels && f < F) v = (float)mel[((long long)b * n_mels + c) * F + f];
    tile[r][tx] = v;
  }
  __syncthreads();
  for (int r = ty; r < 32; r += 8) {
    const int f = f0 + r, c = c0 + tx;
    if (f < F && c < C) melT[((long long)b * (F + 2) + 1 + f) * C + c] = (TD)tile[tx][r];
  }
  // pad rows 0 and F+1
  if (blockIdx.x == 0) {
    for (int i = threadIdx.x; i < 32; i += 256) {
      const int c = c0 + i;
      if (c < C) {
        melT[((long long)b * (F + 2)) * C + c] = (TD)0.f;
        melT[((long long)b * (F + 2) + F + 1) * C + c] = (TD)0.f;
      }
    }
  }
}

template <typename TD, typename TS>
__global__ void convert_kernel(const TS* __restrict__ src, TD* __restrict__ dst, long long n, float scale) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (; i < n; i += stride) dst[i] = (TD)((float)src[i] * scale);
}

template <typename TD, typename TS>
__global__ void conv_weight_reorder_kernel(const TS* __restrict__ src, TD* __restrict__ dst, int co, int ci, int C) {
  // dst[o][k][c] = src[o][c][k]
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long n = (long long)co * 3 * C;
  if (i >= n) return;
  const int c = (int)(i % C);
  const int k = (int)((i / C) % 3);
  const int o = (int)(i / (3LL * C));
  float v = 0.f;
  if (c < ci) v = (float)src[((long long)o * ci + c) * 3 + k];
  dst[i] = (TD)v;
}

template <typename TD, typename TS>
__global__ void interp_positions_kernel(const TS* __restrict__ src, TD* __restrict__ dst, int n_old, int n_new, int d) {
  // torch upsample_linear1d, align_corners=False: src = scale*(dst+0.5)-0.5 clamped at 0 (all in f32)
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long long)n_new * d) return;
  const int c = (int)(i % d);
  const int t = (int)(i / d);
  if (n_new == n_old) { dst[i] = (TD)(float)src[i]; return; }
  const float scale = (float)n_old / (float)n_new;
  float s = scale * ((float)t + 0.5f) - 0.5f;
  if (s < 0.f) s = 0.f;
  int i0 = (int)floorf(s);
  if (i0 > n_old - 1) i0 = n_old - 1;
  const int i1 = (i0 + 1 < n_old) ? i0 + 1 : n_old - 1;
  const float l1 = s - (float)i0;
  const float l0 = 1.0f - l1;
  const float v = l0 * (float)src[(long long)i0 * d + c] + l1 * (float)src[(long long)i1 * d + c];
  dst[i] = (TD)v;
}

template <typename T>
__global__ void embed_kernel(const int* __restrict__ ids, const DecState* __restrict__ stt, const T* __restrict__ tok,
                             const T* __restrict__ pos, T* __restrict__ x, int B, int d, int rows_streams) {
  const int b = blockIdx.x;
  const int id = ids[b];
  int stream, p;
  tw_row_of(b, rows_streams, stt->pos, stream, p);
  for (int i = threadIdx.x; i < d; i += blockDim.x)  // decoder activations are fragment-major (tw_common.h: tw_xt_index)
    x[(long long)(b >> 4) * 16 * d + tw_xt_index<T>(b & 15, i)] = (T)((float)tok[(long long)id * d + i] + (float)pos[(long long)p * d + i]);
}


}  // namespace

hipError_t launch_layernorm(int dtype, const void* x, const void* g, const void* b, void* y, int rows, int d,
                            hipStream_t st) {
  if (rows <= 0) return hipSuccess;
  dim3 grid((rows + 3) / 4);
  const int E = dtype == 0 ? 4 : 8;
  if (d % E != 0 || d > 64 * 5 * E) return hipErrorInvalidValue;
  TW_DISPATCH3(dtype, T, hipLaunchKernelGGL(layernorm_kernel<T>, grid, dim3(256), 0, st, (const T*)x, (const T*)g, (const T*)b, (T*)y, rows, d));
  return hipGetLastError();
}

// two-type kernels: destination = a context element type, source = a caller element type (0 f32, 1 bf16, 2 f16)
#define TW_DISPATCH_DS(dd, sd, TD, TS, ...) TW_DISPATCH3(dd, TD, TW_DISPATCH3(sd, TS, __VA_ARGS__))

hipError_t launch_mel_transpose(int dd, int sd, const void* mel, void* melT, int B, int n_mels, int F, int C,
                                hipStream_t st) {
  if (dd < 0 || dd > 2 || sd < 0 || sd > 2) return hipErrorInvalidValue;
  dim3 grid((F + 31) / 32, (C + 31) / 32, B);
  TW_DISPATCH_DS(dd, sd, TD, TS, hipLaunchKernelGGL((mel_transpose_kernel<TD, TS>), grid, dim3(256), 0, st, (const TS*)mel, (TD*)melT, B, n_mels, F, C));
  return hipGetLastError();
}

hipError_t launch_convert(int dd, int sd, const void* src, void* dst, long long n, float scale, hipStream_t st) {
  if (n <= 0) return hipSuccess;
  if (dd < 0 || dd > 2 || sd < 0 || sd > 2) return hipErrorInvalidValue;
  long long blocks = (n + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  dim3 grid((unsigned)blocks);
  TW_DISPATCH_DS(dd, sd, TD, TS, hipLaunchKernelGGL((convert_kernel<TD, TS>), grid, dim3(256), 0, st, (const TS*)src, (TD*)dst, n, scale));
  return hipGetLastError();
}

hipError_t launch_fill_zero(void* dst, long long bytes, hipStream_t st) {
  if (bytes <= 0) return hipSuccess;
  return hipMemsetAsync(dst, 0, (size_t)bytes, st);
}

hipError_t launch_conv_weight_reorder(int dd, int sd, const void* src, void* dst, int co, int ci, int C,
                                      hipStream_t st) {
  if (dd < 0 || dd > 2 || sd < 0 || sd > 2) return hipErrorInvalidValue;
  const long long n = (long long)co * 3 * C;
  dim3 grid((unsigned)((n + 255) / 256));
  TW_DISPATCH_DS(dd, sd, TD, TS, hipLaunchKernelGGL((conv_weight_reorder_kernel<TD, TS>), grid, dim3(256), 0, st, (const TS*)src, (TD*)dst, co, ci, C));
  return hipGetLastError();
}

hipError_t launch_interp_positions(int dd, int sd, const void* src, void* dst, int n_old, int n_new, int d,
                                   hipStream_t st) {
  if (sd != 0 || dd < 0 || dd > 2) return hipErrorInvalidValue;
  const long long n = (long long)n_new * d;
  dim3 grid((unsigned)((n + 255) / 256));
  TW_DISPATCH3(dd, TD, hipLaunchKernelGGL((interp_positions_kernel<TD, float>), grid, dim3(256), 0, st, (const float*)src, (TD*)dst, n_old, n_new, d));
  return hipGetLastError();
}

hipError_t launch_embed(int dtype, const int* ids, const DecState* stt, const void* tok, const void* pos, void* x,
                        int B, int d, int rows_streams, hipStream_t st) {
  TW_DISPATCH3(dtype, T, hipLaunchKernelGGL(embed_kernel<T>, dim3(B), dim3(256), 0, st, ids, stt, (const T*)tok, (const T*)pos, (T*)x, B, d, rows_streams));
  return hipGetLastError();
}
