// MFMA GEMM for the MFMA-bound part of the hot path (A2 conv stem, A3 encoder projections/FFN,
// A5 cross-K/V projection):  C[M,N] = epilogue(A[M,K] . W[N,K]^T),  both operands K-contiguous
// (activations token-major, weights in the HF nn.Linear [out,in] layout).
//
// gfx950 design
//  * 256 threads = 4 waves in a 2x2 arrangement; block tile BM (activation rows) x BN (weight rows),
//    K tile = 8 x 16 B per row (64 bf16 / 32 f32).  128x128 for large M, 64x64 when the grid
//    would otherwise not cover the 256 CUs.
//  * LDS image: [row][k-slot ^ ((row>>1)&7)] of 16-B vectors, filled by direct global->LDS DMA
//    (global_load_lds_dwordx4, swizzle applied on the source side: 8 lanes still cover one 128-B
//    row segment); the MFMA fragment read (16 lanes = 16 consecutive rows of one k-slot,
//    ds_read_b128) touches every bank once.
//  * "swapped" MFMA: the weight tile is the A operand and the activation tile the B operand of
//    v_mfma_f32_16x16x32_bf16 (or 4 x v_mfma_f32_16x16x4_f32 in strict-f32 mode, using a
//    k-permutation so that the same 16-B fragment feeds both), so every lane ends up with 4
//    consecutive output columns of one row -> 8/16-B epilogue stores and vector bias loads.
//  * double-buffered LDS, one barrier per K tile: the DMA of tile k+1 is issued right after the
//    barrier that retires tile k and runs under the MFMAs of tile k (no staging registers, no
//    ds_write pass: 1.2-1.5x the register-staged version of this kernel).
//  * fused epilogues: bias, exact GELU, residual / positional add, and the head-split / transposed
//    layouts the attention kernels consume (no separate permute kernels).
#include "tw_common.h"

#include <cstdlib>

namespace {

__device__ __forceinline__ long long rowmap(const RowMap& r, int m) {
  return (long long)(m / r.rpb) * r.bstride + (long long)(m % r.rpb) * r.rstride;
}

// erf-GELU (activation_function = "gelu").  Strict-f32 contexts use the library erff; bf16 contexts, whose outputs are
// rounded to 8 mantissa bits anyway, use the Abramowitz-Stegun 7.1.26 rational form (|error| < 2e-7 on erf + one fast
// exp: ~12 instead of ~45 VALU instructions per element, which is a third of the fc1 GEMM's epilogue-bound run time).
template <typename T>
__device__ __forceinline__ float gelu_exact(float x) {
  const float z = x * 0.70710678118654752440f;
  if (sizeof(T) == 4) return 0.5f * x * (1.0f + erff(z));
  const float az = fabsf(z);
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, az, 1.0f));
  float p = fmaf(t, 1.061405429f, -1.453152027f);
  p = fmaf(t, p, 1.421413741f);
  p = fmaf(t, p, -0.284496736f);
  p = fmaf(t, p, 0.254829592f);
  const float e = 1.0f - p * t * __expf(-az * az);
  return 0.5f * x * (1.0f + copysignf(e, z));
}

template <typename T> struct Vec4;  // 4 consecutive elements
template <> struct Vec4<float> {
  float4 v;
  __device__ void load(const float* p) { v = *reinterpret_cast<const float4*>(p); }
  __device__ void store(float* p) const { *reinterpret_cast<float4*>(p) = v; }
  __device__ float get(int i) const { return i == 0 ? v.x : i == 1 ? v.y : i == 2 ? v.z : v.w; }
  __device__ void set(int i, float f) { if (i == 0) v.x = f; else if (i == 1) v.y = f; else if (i == 2) v.z = f; else v.w = f; }
  __device__ float elem(int i) const { return get(i); }
};
template <> struct Vec4<bf16_t> {
  bf16_t e[4];
  __device__ void load(const bf16_t* p) { *reinterpret_cast<uint2*>(e) = *reinterpret_cast<const uint2*>(p); }
  __device__ void store(bf16_t* p) const { *reinterpret_cast<uint2*>(p) = *reinterpret_cast<const uint2*>(e); }
  __device__ float get(int i) const { return (float)e[i]; }
  __device__ void set(int i, float f) { e[i] = (bf16_t)f; }
  __device__ bf16_t elem(int i) const { return e[i]; }
};

template <typename T>
__device__ __forceinline__ f32x4_t mfma_step(const u32x4_t& wfrag, const u32x4_t& afrag, f32x4_t acc);

template <>
__device__ __forceinline__ f32x4_t mfma_step<bf16_t>(const u32x4_t& wfrag, const u32x4_t& afrag, f32x4_t acc) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, wfrag),
                                                 __builtin_bit_cast(bf16x8_t, afrag), acc, 0, 0, 0);
}
template <>
__device__ __forceinline__ f32x4_t mfma_step<float>(const u32x4_t& wfrag, const u32x4_t& afrag, f32x4_t acc) {
  const f32x4_t w = __builtin_bit_cast(f32x4_t, wfrag);
  const f32x4_t a = __builtin_bit_cast(f32x4_t, afrag);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w[0], a[0], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w[1], a[1], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w[2], a[2], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w[3], a[3], acc, 0, 0, 0);
  return acc;
}

// BM x BN block tile, WM x WN wavefronts (each owns a (BM/WM) x (BN/WN) sub-tile), ST-stage LDS ring filled by global->LDS
// DMA with prefetch distance ST-1: with two stages the DMA of tile k+1 has one tile's worth of MFMAs (~500 cycles) to land,
// less than an HBM round trip, and the loop runs at a fifth of the matrix-core rate; with three stages (distance 2, 8
// wavefronts = 2 per SIMD on a 256 x 128 tile: ~1000 cycles of MFMAs per SIMD and iteration) the latency is covered.
// One barrier per K tile: after it, tile kt is visible to every wavefront and buffer (kt-1) % ST - consumed in the
// previous iteration - is free for tile kt + ST - 1.
template <typename T, int BM, int BN, int WM, int WN, int ST>
__global__ __launch_bounds__(WM * WN * 64) void gemm_kernel(const T* __restrict__ A, RowMap amap, const T* __restrict__ W,
                                                              int M, int N, int K, GemmEpilogue ep) {
  constexpr int E = ElemTraits<T>::kPer16B;  // elements per 16-B vector
  constexpr int BKE = 8 * E;                 // elements per K tile
  constexpr int NWAVE = WM * WN;
  constexpr int NTHR = NWAVE * 64;
  constexpr int AV = BM * 8 / NTHR;          // 16-B vectors per thread per tile (activations)
  constexpr int WV = BN * 8 / NTHR;          // (weights)
  constexpr int RM = BM / WM, RN = BN / WN;  // rows / columns of a wavefront's sub-tile
  constexpr int MT = RM / 16;                // 16-row MFMA tiles per wave along M
  constexpr int NT = RN / 16;                // along N
  static_assert(AV >= 1 && WV >= 1 && MT >= 1 && NT >= 1, "tile / wavefront layout");
  constexpr int STAGE = 8 * (BM + BN);       // 16-B vectors per ring stage
  __shared__ u32x4_t lds[ST * STAGE];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wn = wave % WN;  // wave position along N
  const int wm = wave / WN;  // along M
  const int n0 = blockIdx.x * BN;
  const int m0 = blockIdx.y * BM;

  // Global -> LDS by direct DMA (global_load_lds_dwordx4): no staging registers, no ds_write pass.  The DMA writes the 64
  // lanes of a wavefront to 64 consecutive 16-B LDS slots, so the swizzle is applied on the SOURCE side: linear LDS
  // position p (16-B units) of a tile holds row p>>3, k-slot (p&7) ^ ((row>>1)&7); 8 consecutive lanes still cover one
  // 128-B row segment of global memory (coalesced), and the fragment read - 16 lanes = 16 consecutive rows of one k-slot,
  // ds_read_b128 - touches every bank exactly once.
  const T* asrc[AV];
#pragma unroll
  for (int i = 0; i < AV; ++i) {
    const int p = (i * NWAVE + wave) * 64 + lane;
    const int row = p >> 3, slot = (p & 7) ^ ((row >> 1) & 7);
    int m = m0 + row;
    if (m >= M) m = M - 1;
    asrc[i] = A + rowmap(amap, m) + slot * E;
  }
  const T* wsrc[WV];
#pragma unroll
  for (int i = 0; i < WV; ++i) {
    const int p = (i * NWAVE + wave) * 64 + lane;
    const int row = p >> 3, slot = (p & 7) ^ ((row >> 1) & 7);
    wsrc[i] = W + (long long)min(n0 + row, N - 1) * K + slot * E;
  }
  typedef __attribute__((address_space(1))) const void* gptr_t;
  typedef __attribute__((address_space(3))) void* lptr_t;
  auto issue_tile = [&](int kt, int b) {
    const int koff = kt * BKE;
    u32x4_t* sa = lds + b * STAGE;
    u32x4_t* sw = sa + 8 * BM;
#pragma unroll
    for (int i = 0; i < AV; ++i)
      __builtin_amdgcn_global_load_lds((gptr_t)(asrc[i] + koff), (lptr_t)(sa + (i * NWAVE + wave) * 64), 16, 0, 0);
#pragma unroll
    for (int i = 0; i < WV; ++i)
      __builtin_amdgcn_global_load_lds((gptr_t)(wsrc[i] + koff), (lptr_t)(sw + (i * NWAVE + wave) * 64), 16, 0, 0);
  };

  f32x4_t acc[NT][MT];
#pragma unroll
  for (int a = 0; a < NT; ++a)
#pragma unroll
    for (int b = 0; b < MT; ++b) acc[a][b] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  const int nk = K / BKE;
  const int fr = lane & 15;  // fragment row within a 16-row tile
  const int fq = lane >> 4;  // k-slot quad
#pragma unroll
  for (int t = 0; t < ST - 1; ++t)
    if (t < nk) issue_tile(t, t);
  int buf = 0;
  for (int kt = 0; kt < nk; ++kt) {
    // tile kt has landed: this wavefront's own DMA by vmcnt (requests retire in order; the ST-2 younger tiles may stay in
    // flight), the other wavefronts' by the barrier
    if (ST > 2 && kt + ST - 2 < nk) {
      if constexpr (ST == 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(AV + WV) : "memory");
      else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * (AV + WV)) : "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    if (kt + ST - 1 < nk) {
      int nb = buf + ST - 1;
      if (nb >= ST) nb -= ST;
      issue_tile(kt + ST - 1, nb);
    }
    const u32x4_t* la = lds + buf * STAGE;
    const u32x4_t* lw = la + 8 * BM;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const int slot = kk * 4 + fq;
      u32x4_t af[MT], wf[NT];
#pragma unroll
      for (int b = 0; b < MT; ++b) {
        const int row = wm * RM + b * 16 + fr;
        af[b] = la[row * 8 + (slot ^ ((row >> 1) & 7))];
      }
#pragma unroll
      for (int a = 0; a < NT; ++a) {
        const int row = wn * RN + a * 16 + fr;
        wf[a] = lw[row * 8 + (slot ^ ((row >> 1) & 7))];
      }
#pragma unroll
      for (int a = 0; a < NT; ++a)
#pragma unroll
        for (int b = 0; b < MT; ++b) acc[a][b] = mfma_step<T>(wf[a], af[b], acc[a][b]);
    }
    if (++buf == ST) buf = 0;
  }

  // ---- epilogue: lane holds, per (a,b) tile, 4 consecutive columns n of row m ----
  const T* bias = reinterpret_cast<const T*>(ep.bias);
  const T* res = reinterpret_cast<const T*>(ep.res);
  const int dmodel = ep.H * 64;
#pragma unroll
  for (int b = 0; b < MT; ++b) {
    const int m = m0 + wm * RM + b * 16 + fr;
    if (m >= M) continue;
    long long roff = 0;
    if (res) roff = rowmap(ep.res_map, ep.res_mod > 0 ? (m % ep.res_mod) : m);
#pragma unroll
    for (int a = 0; a < NT; ++a) {
      const int n = n0 + wn * RN + a * 16 + fq * 4;
      if (n >= N) continue;
      float v[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = acc[a][b][r];
      if (bias) {
        Vec4<T> bv;
        bv.load(bias + n);
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] += bv.get(r);
      }
      if (ep.gelu) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = gelu_exact<T>(v[r]);
      }
      if (res) {
        Vec4<T> rv;
        rv.load(res + roff + n);
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] += rv.get(r);
      }
      Vec4<T> ov;
#pragma unroll
      for (int r = 0; r < 4; ++r) ov.set(r, v[r]);
      if (ep.mode == EPI_ROWMAJOR) {
        ov.store(reinterpret_cast<T*>(ep.out) + rowmap(ep.c_map, m) + n);
      } else {
        const int bidx = m / ep.T, t = m % ep.T;
        int seg = n / dmodel;
        const int nn = n - seg * dmodel;
        const int h = nn >> 6, dd = nn & 63;
        if (ep.mode == EPI_HEADSPLIT) seg = 0;
        if (ep.mode == EPI_QKV_ENC && seg == 2) {
          T* o = reinterpret_cast<T*>(ep.out3) + ((long long)(bidx * ep.H + h) * 64 + dd) * ep.Tp + t;
#pragma unroll
          for (int r = 0; r < 4; ++r) o[(long long)r * ep.Tp] = ov.elem(r);
        } else if (ep.mode == EPI_KV_CROSS) {
          // decoder cross-attention operands, fragment-major per (stream, head) with Tp keys (tw_common.h)
          const long long hb = (long long)(bidx * ep.H + h) * ep.Tp * 64;
          if (seg == 0) {
            ov.store(reinterpret_cast<T*>(ep.out) + hb + tw_kf_index<T>(t, dd));  // 4 dims of one key: one 16-B vector
          } else {
            T* o = reinterpret_cast<T*>(ep.out2) + hb;
#pragma unroll
            for (int r = 0; r < 4; ++r) o[tw_vtf_index<T>(t, dd + r)] = ov.elem(r);
          }
        } else {
          T* base = reinterpret_cast<T*>(seg == 0 ? ep.out : ep.out2);
          ov.store(base + ((long long)(bidx * ep.H + h) * ep.T + t) * 64 + dd);
        }
      }
    }
  }
}

}  // namespace

static int gemm_env(const char* name, int dflt) {
  const char* e = getenv(name);
  return e ? atoi(e) : dflt;
}

template <typename T, int BM, int BN, int WM, int WN, int ST>
static hipError_t gemm_go(const void* A, RowMap amap, const void* W, int M, int N, int K, const GemmEpilogue& ep, hipStream_t st) {
  dim3 grid((N + BN - 1) / BN, (M + BM - 1) / BM);
  hipLaunchKernelGGL((gemm_kernel<T, BM, BN, WM, WN, ST>), grid, dim3(WM * WN * 64), 0, st, reinterpret_cast<const T*>(A), amap,
                     reinterpret_cast<const T*>(W), M, N, K, ep);
  return hipGetLastError();
}

template <typename T>
static hipError_t gemm_dispatch(const void* A, RowMap amap, const void* W, int M, int N, int K,
                                const GemmEpilogue& ep, hipStream_t st) {
  constexpr int E = ElemTraits<T>::kPer16B;
  if (M <= 0) return hipSuccess;
  if (K % (8 * E) != 0 || N % 64 != 0) return hipErrorInvalidValue;
  // Tile choice by how many workgroups cover the 256 compute units (TW_GEMM_CFG forces one for experiments):
  //   3: 256 x 128, 8 wavefronts, 3-stage ring   - large M (batched encoder): halves the LDS fill per flop
  //   2: 128 x 128, 4 wavefronts, 3-stage ring
  //   1: 128 x 64, 4 wavefronts, 3 stages        - mid-size (single 30 s stream): >= 1 workgroup per CU with 2x the MFMAs per
  //                                                LDS byte of the 64 x 64 tile
  //   0: 64 x 64, 4 wavefronts, 2 stages         - small M
  static const int forced = gemm_env("TW_GEMM_CFG", -1);
  const long long b256 = (long long)((M + 255) / 256) * ((N + 127) / 128);
  const long long b128 = (long long)((M + 127) / 128) * ((N + 127) / 128);
  const long long b12864 = (long long)((M + 127) / 128) * (N / 64);
  int cfg;
  if (forced >= 0) cfg = forced;
  else if (N % 128 == 0 && b256 >= 512) cfg = 3;
  else if (N % 128 == 0 && b128 >= 384) cfg = 2;
  else if (b12864 >= 200) cfg = 1;
  else cfg = 0;
  if (sizeof(T) == 4 && cfg == 3) cfg = 2;  // strict-f32 contexts: parity mode, the 8-wavefront tile is not instantiated
  switch (cfg) {
    case 3:
      if constexpr (sizeof(T) == 2) return gemm_go<T, 256, 128, 4, 2, 3>(A, amap, W, M, N, K, ep, st);
      return hipErrorInvalidValue;
    case 2: return gemm_go<T, 128, 128, 2, 2, 3>(A, amap, W, M, N, K, ep, st);
    case 4: return gemm_go<T, 128, 128, 2, 2, 2>(A, amap, W, M, N, K, ep, st);   // round-1 kernel (A/B runs)
    case 1: return gemm_go<T, 128, 64, 2, 2, 3>(A, amap, W, M, N, K, ep, st);
    default: return gemm_go<T, 64, 64, 2, 2, 2>(A, amap, W, M, N, K, ep, st);
  }
}

hipError_t launch_gemm(int dtype, const void* A, RowMap amap, const void* W, int M, int N, int K,
                       const GemmEpilogue& ep, hipStream_t st) {
  if (dtype == 1) return gemm_dispatch<bf16_t>(A, amap, W, M, N, K, ep, st);
  return gemm_dispatch<float>(A, amap, W, M, N, K, ep, st);
}
