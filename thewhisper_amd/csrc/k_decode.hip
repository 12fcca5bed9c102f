// A6-A8, A10: the per-token decoder path.  With one new token per stream every projection is a
// skinny GEMM  y[B<=16, N] = f(x[B, K]) . W[N, K]^T  whose cost is streaming W once from HBM
// (1.6 GB per step for large-v3 in bf16), so these kernels are HBM-bound by construction:
//
//  * gemv_kernel: one wavefront owns R output rows for the FULL K (no split-K: deterministic sums,
//    fused pre-LayerNorm / bias / GELU / residual / KV-cache scatter).  Lanes stride along K with
//    16-byte loads (1 KiB contiguous per wave instruction); the <=16 activation vectors are staged
//    once per workgroup in LDS (after the fused LayerNorm) and every weight vector is reused for
//    all B streams with v_dot2c_f32_bf16 (bf16 mode) or FMAs (strict-f32 mode).
//  * dec_self_attn / dec_cross_attn: single-query attention over the KV cache, one wavefront per
//    (stream, head) [x4 for the 500..1500-key cross attention], lanes = 4 key groups x 16 dim
//    quads, online softmax in registers, 16-lane shuffle dot products.  Cross attention also emits
//    the softmax rows of the alignment heads for the DTW (A11).
//  * sampler: Whisper's logits processors + first-index argmax on the fp32 logits, one workgroup
//    per stream (HF:generation/logits_process.py:1816-2047, HF:generation/utils.py:2925).
#include "tw_common.h"

#include <cstdlib>

namespace {

__device__ __forceinline__ float row16_sum(float v) { return tw_row16_sum(v); }
__device__ __forceinline__ float wave_sum(float v) { return tw_wave_sum(v); }
__device__ __forceinline__ float gelu_exact(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }

template <typename T> __device__ __forceinline__ float dot16(const u32x4_t& w, const u32x4_t& x, float acc);
template <> __device__ __forceinline__ float dot16<bf16_t>(const u32x4_t& w, const u32x4_t& x, float acc) {
  // NB: written without a loop over w[i]: hipcc (ROCm 7.2) folded `bit_cast<bf16x2>(w[i])` in an unrolled
  // loop to element 0 for every i (seen in the ISA: four identical v_dot2c), so the pairs are named.
  const bf16x8_t a = __builtin_bit_cast(bf16x8_t, w), b = __builtin_bit_cast(bf16x8_t, x);
  acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(a, a, 0, 1), __builtin_shufflevector(b, b, 0, 1), acc, false);
  acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(a, a, 2, 3), __builtin_shufflevector(b, b, 2, 3), acc, false);
  acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(a, a, 4, 5), __builtin_shufflevector(b, b, 4, 5), acc, false);
  acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(a, a, 6, 7), __builtin_shufflevector(b, b, 6, 7), acc, false);
  return acc;
}
template <> __device__ __forceinline__ float dot16<float>(const u32x4_t& w, const u32x4_t& x, float acc) {
  const f32x4_t a = __builtin_bit_cast(f32x4_t, w), b = __builtin_bit_cast(f32x4_t, x);
#pragma unroll
  for (int i = 0; i < 4; ++i) acc = fmaf(a[i], b[i], acc);
  return acc;
}

template <typename T> __device__ __forceinline__ void unpack16(const u32x4_t& v, float* out);
template <> __device__ __forceinline__ void unpack16<float>(const u32x4_t& v, float* out) {
  const f32x4_t f = __builtin_bit_cast(f32x4_t, v);
  out[0] = f[0]; out[1] = f[1]; out[2] = f[2]; out[3] = f[3];
}
template <> __device__ __forceinline__ void unpack16<bf16_t>(const u32x4_t& v, float* out) {
  const bf16x8_t f = __builtin_bit_cast(bf16x8_t, v);
#pragma unroll
  for (int i = 0; i < 8; ++i) out[i] = (float)f[i];
}
template <typename T> __device__ __forceinline__ u32x4_t pack16(const float* in);
template <> __device__ __forceinline__ u32x4_t pack16<float>(const float* in) {
  return __builtin_bit_cast(u32x4_t, f32x4_t{in[0], in[1], in[2], in[3]});
}
template <> __device__ __forceinline__ u32x4_t pack16<bf16_t>(const float* in) {
  bf16x8_t f;
#pragma unroll
  for (int i = 0; i < 8; ++i) f[i] = (bf16_t)in[i];
  return __builtin_bit_cast(u32x4_t, f);
}

constexpr int GEMV_LDS_BUDGET = 80 * 1024;  // activations staged per workgroup (2 workgroups / CU)
constexpr int GEMV_D = 4;                    // weight vectors in flight per lane and row (register ring)
constexpr int GEMV_RG = 8;                   // row groups per workgroup for very tall matrices (logits)

template <typename T, int BT> struct GemvChunk {
  static constexpr int E = ElemTraits<T>::kPer16B;
  // largest K chunk (in 16-B vectors, multiple of 64) whose BT activation rows fit the LDS budget
  static constexpr int kVec = (GEMV_LDS_BUDGET / (BT * 16) / 64) * 64;
};

// Folded pre-LayerNorm.  For y = LN(x) . W^T + bias with LN(x) = (x - mean) * rstd * g + beta:
//     y[n] = rstd * ( x . W'[n]  -  mean * gW[n] )  +  cb[n],
//     W'[n,k] = g[k] W[n,k],   gW[n] = sum_k g[k] W[n,k],   cb[n] = sum_k beta[k] W[n,k] + bias[n]
// W', gW and cb are prepared once at weight-finalisation time (fold_ln_kernel below), so the projection streams W'
// against the RAW residual stream exactly like a projection without LayerNorm, and only the two per-stream scalars
// (mean, rstd) have to be known when the epilogue runs - they are computed off the critical path while the weight
// loads are in flight.  No normalised copy of x is ever materialised.

// y[b, n] = epi( x[b, :] . W[n, :] )   one wavefront = R rows x full K, lanes stride K by 16 B  (1..4 streams).
// Latency structure (the decode step is a chain of ~260 such launches, so this matters as much as bandwidth): the
// first GEMV_D weight vectors of every row and the epilogue operands are requested BEFORE the activations are staged,
// the staging loads are issued in batches of 8 per thread, LayerNorm statistics are reduced between the staging
// barrier and the (already in flight) weight data, and the wave reduction uses DPP/permlane steps only.
template <typename T, int BT, int R, bool MULTI>
__global__ __launch_bounds__(256) void gemv_kernel(GemvArgs a) {
  constexpr int E = ElemTraits<T>::kPer16B;
  constexpr int D = GEMV_D;
  constexpr int CV = GemvChunk<T, BT>::kVec;  // vectors per chunk in chunked mode
  constexpr int IPC = CV / 64;
  constexpr int MAXV = (1280 / E + 63) / 64;  // LayerNorm rows have K <= 1280
  constexpr int RG = MULTI ? GEMV_RG : 1;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  __shared__ float stat[16][2];               // (mean, rstd) per stream
  u32x4_t* xs = reinterpret_cast<u32x4_t*>(smem);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int K = a.K, N = a.N, B = a.B;
  const T* x = reinterpret_cast<const T*>(a.x);
  const T* W = reinterpret_cast<const T*>(a.W);
  const int nv_row = K / E;
  const int n_it = (nv_row + 63) / 64;
  const bool chunked = nv_row > CV;
  const int xld = chunked ? CV : nv_row;  // LDS row stride in vectors
  const bool has_ln = a.ln_gw != nullptr;

  int row0 = ((blockIdx.x * RG) * 4 + wave) * R;
  const T* wrow[R];
  u32x4_t wq[D][R];
  auto prime = [&]() {
#pragma unroll
    for (int r = 0; r < R; ++r) wrow[r] = W + (long long)min(row0 + r, N - 1) * K;
#pragma unroll
    for (int j = 0; j < D; ++j) {
      const int vi = j * 64 + lane;
#pragma unroll
      for (int r = 0; r < R; ++r)  // unconditional (clamped): lanes past the row end re-read the last vector, unused
        wq[j][r] = *reinterpret_cast<const u32x4_t*>(wrow[r] + (long long)min(vi, nv_row - 1) * E);
    }
  };
  // NB on ordering: the vector-memory counter retires loads IN ORDER and a load inside a conditional block gets its own
  // wait, so every operand is requested unconditionally (clamped addresses, dummy pointers for absent operands) and in
  // the order it is needed: epilogue scalars, then the activation staging loads, and only then the HBM weight prefetch
  // (otherwise the LDS staging store would have to wait for the whole weight prefetch as well).
  const T* bias = reinterpret_cast<const T*>(a.bias);
  const T* res = reinterpret_cast<const T*>(a.res);
  const float* gw_p = has_ln ? a.ln_gw : reinterpret_cast<const float*>(W);
  const float* cb_p = has_ln ? a.ln_cb : reinterpret_cast<const float*>(W);
  const T* bias_p = bias ? bias : W;
  const T* res_p = res ? res : W;
  const DecState* stt_p = a.stt ? a.stt : reinterpret_cast<const DecState*>(W);
  const int cur_pos = stt_p->pos;  // only meaningful when a.stt != null (KV-cache scatter)
  auto load_epi = [&](int row0_, float (&e_c)[R], float (&e_gw)[R], float (&e_res)[R]) {
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int n = min(row0_ + r, N - 1);
      const float v_gw = gw_p[has_ln ? n : 0];
      const float v_cb = cb_p[has_ln ? n : 0];
      const float v_b = (float)bias_p[bias ? n : 0];
      const float v_r = (float)res_p[res ? (long long)min(lane, B - 1) * a.ldres + n : 0];
      e_gw[r] = v_gw;
      e_c[r] = has_ln ? v_cb : (bias ? v_b : 0.f);
      e_res[r] = res ? v_r : 0.f;
    }
  };
  float e_c0[R], e_gw0[R], e_res0[R];
  load_epi(row0, e_c0, e_gw0, e_res0);
  // stage `cv` vectors per activation row starting at vector `cbase`; loads issued 8 per thread at a time
  constexpr int NB = BT <= 4 ? 4 : 8;  // staging loads per thread and batch (one batch covers K <= 8192/BT... see launcher)
  auto stage_load = [&](int cbase, int cv, int i0, u32x4_t (&tmp)[NB]) {
    const int total = BT * cv;
#pragma unroll
    for (int u = 0; u < NB; ++u) {
      const int i = i0 + u * 256;
      const int b = min(i / cv, BT - 1), vi = i - (i / cv) * cv;
      const bool on = (i < total) && (i / cv) < B;
      const T* src = x + (long long)min(b, B - 1) * a.ldx + (long long)(cbase + vi) * E;
      const u32x4_t v = *reinterpret_cast<const u32x4_t*>(src);  // unconditional (clamped) so the loads batch
      tmp[u] = on ? v : u32x4_t{0u, 0u, 0u, 0u};
    }
  };
  auto stage_store = [&](int cv, int i0, const u32x4_t (&tmp)[NB]) {
    const int total = BT * cv;
#pragma unroll
    for (int u = 0; u < NB; ++u) {
      const int i = i0 + u * 256;
      if (i < total) { const int b = i / cv, vi = i - b * cv; xs[b * xld + vi] = tmp[u]; }
    }
  };
  auto stage = [&](int cbase, int cv) {
    const int total = BT * cv;
    for (int i0 = tid; i0 < total; i0 += 256 * NB) {
      u32x4_t tmp[NB];
      stage_load(cbase, cv, i0, tmp);
      stage_store(cv, i0, tmp);
    }
  };

  if (!chunked) {
    u32x4_t first[NB];
    stage_load(0, nv_row, tid, first);   // first (usually only) staging batch: requested before the weights
    prime();
    stage_store(nv_row, tid, first);
    for (int i0 = tid + 256 * NB; i0 < BT * nv_row; i0 += 256 * NB) {
      u32x4_t tmp[NB];
      stage_load(0, nv_row, i0, tmp);
      stage_store(nv_row, i0, tmp);
    }
    __syncthreads();
    if (has_ln) {  // statistics of the raw rows (two-pass, eps 1e-5, biased variance); wave w owns streams w, w+4, ..
      for (int b = wave; b < B; b += 4) {
        float v[MAXV][E];
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < MAXV; ++i) {
          const int vi = lane + i * 64;
          if (vi < nv_row) {
            unpack16<T>(xs[b * xld + vi], v[i]);
#pragma unroll
            for (int e = 0; e < E; ++e) s += v[i][e];
          }
        }
        const float mean = wave_sum(s) / (float)K;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < MAXV; ++i) {
          const int vi = lane + i * 64;
          if (vi < nv_row) {
#pragma unroll
            for (int e = 0; e < E; ++e) { const float c = v[i][e] - mean; q += c * c; }
          }
        }
        const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)K + 1e-5f);
        if (lane == 0) { stat[b][0] = mean; stat[b][1] = rstd; }
      }
    }
  } else {
    prime();
  }

  for (int grp = 0; grp < RG; ++grp) {
    if (grp > 0) {
      row0 = ((blockIdx.x * RG + grp) * 4 + wave) * R;
      prime();
    }
    float e_c[R], e_gw[R], e_res[R];
    if (grp == 0) {
#pragma unroll
      for (int r = 0; r < R; ++r) { e_c[r] = e_c0[r]; e_gw[r] = e_gw0[r]; e_res[r] = e_res0[r]; }
    } else {
      load_epi(row0, e_c, e_gw, e_res);
    }
    float acc[R][BT];
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
      for (int b = 0; b < BT; ++b) acc[r][b] = 0.f;

    for (int it0 = 0; it0 < n_it; it0 += D) {
#pragma unroll
      for (int j = 0; j < D; ++j) {
        const int it = it0 + j;
        if (it < n_it) {
          int cbase = 0;
          if (chunked) {
            cbase = (it / IPC) * CV;
            if (it % IPC == 0) {
              if (it > 0) __syncthreads();
              stage(cbase, min(CV, nv_row - cbase));
              __syncthreads();
            }
          }
          const int vi = it * 64 + lane;
          if (vi < nv_row) {
#pragma unroll
            for (int b = 0; b < BT; ++b) {
              const u32x4_t xv = xs[b * xld + (vi - cbase)];
#pragma unroll
              for (int r = 0; r < R; ++r) acc[r][b] = dot16<T>(wq[j][r], xv, acc[r][b]);
            }
          }
          const int vn = (it + D) * 64 + lane;
#pragma unroll
          for (int r = 0; r < R; ++r)
            wq[j][r] = *reinterpret_cast<const u32x4_t*>(wrow[r] + (long long)min(vn, nv_row - 1) * E);
        }
      }
    }

    // ---- reduce across the wavefront; lane b then owns stream b (one predicated store per row) ----
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
      for (int b = 0; b < BT; ++b) acc[r][b] = wave_sum(acc[r][b]);
    if (has_ln && grp == 0) __syncthreads();  // statistics written by the owning wavefronts are visible
    float mean = 0.f, rstd = 1.f;
    if (has_ln && lane < B) { mean = stat[lane][0]; rstd = stat[lane][1]; }
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int n = row0 + r;
      float mine = 0.f;
#pragma unroll
      for (int b = 0; b < BT; ++b) mine = (lane == b) ? acc[r][b] : mine;
      if (n < N && lane < B) {
        const int b = lane;
        float v = has_ln ? rstd * (mine - mean * e_gw[r]) + e_c[r] : mine + e_c[r];
        if (a.gelu) v = gelu_exact(v);
        v += e_res[r];
        if (a.y_f32) {
          a.y_f32[(long long)b * N + n] = v;
        } else if (a.kcache && n >= a.d_model) {
          const int seg = n / a.d_model;
          T* dst = reinterpret_cast<T*>(seg == 1 ? a.kcache : a.vcache);
          dst[(long long)b * a.cache_bstride + (long long)cur_pos * a.d_model + (n - seg * a.d_model)] = (T)v;
        } else {
          reinterpret_cast<T*>(a.y)[(long long)b * a.ldy + n] = (T)v;
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Skinny MFMA GEMM for 5..16 concurrent streams:  y[b, n0..n0+15] for one 16-row weight tile per
// workgroup, K split over the NW wavefronts.  With 16 streams the activation block is exactly one
// MFMA tile (v_mfma_f32_16x16x32_bf16: A = 16 weight rows x 32 k, B = 16 streams x 32 k), so each
// 1-KiB weight fragment costs ONE matrix instruction and ONE 16-B activation fragment instead of
// 16 LDS reads + 64 dot2 + a wave reduction per stream in the VALU formulation; the contraction
// over k happens inside the MFMA and only NW partial 16x16 tiles are summed through LDS.
// All weight and activation fragments of a wavefront are requested up front (activations straight
// from L2 - the residual stream is 40 KB); with the folded LayerNorm (above) the per-stream
// (mean, rstd) are reduced by the wavefronts while those loads are in flight and meet the
// accumulators at the one barrier the partial-tile reduction needs anyway.
// Deterministic: fixed summation order, no atomics.
// ---------------------------------------------------------------------------------------------

template <typename T>
__device__ __forceinline__ f32x4_t sk_mfma(const u32x4_t& w, const u32x4_t& x, f32x4_t acc);
template <>
__device__ __forceinline__ f32x4_t sk_mfma<bf16_t>(const u32x4_t& w, const u32x4_t& x, f32x4_t acc) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, w), __builtin_bit_cast(bf16x8_t, x), acc, 0, 0, 0);
}
template <>
__device__ __forceinline__ f32x4_t sk_mfma<float>(const u32x4_t& w, const u32x4_t& x, f32x4_t acc) {
  const f32x4_t a = __builtin_bit_cast(f32x4_t, w), b = __builtin_bit_cast(f32x4_t, x);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[0], b[0], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[1], b[1], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[2], b[2], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[3], b[3], acc, 0, 0, 0);
  return acc;
}

// SK_MAXS = weight/activation fragments a wavefront keeps in flight (5 covers K = 1280 in bf16 with 8 wavefronts)
template <typename T, int NW, int SK_MAXS>
__global__ __launch_bounds__(NW * 64, (NW >= 16 ? 4 : (SK_MAXS <= 5 ? 4 : 2))) void skinny_mfma_kernel(GemvArgs a) {
  constexpr int E = ElemTraits<T>::kPer16B;
  constexpr int MAXV = (1280 / E + 63) / 64;
  constexpr int RPW = (16 + NW - 1) / NW;  // LayerNorm-statistics rows per wavefront
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  __shared__ float stat[16][2];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 15, kq = lane >> 4;
  const int RG = a.rg;
  const int K = a.K, N = a.N, B = a.B;
  const T* x = reinterpret_cast<const T*>(a.x);
  const T* W = reinterpret_cast<const T*>(a.W);
  const int nv_row = K / E;          // 16-B vectors per row
  const int S = nv_row / 4;          // 64-B steps per row
  const int s_lo = (int)((long long)wave * S / NW), s_hi = (int)((long long)(wave + 1) * S / NW);
  const bool has_ln = a.ln_gw != nullptr;
  float* red = reinterpret_cast<float*>(smem);  // [NW][256]

  int n0 = blockIdx.x * RG * 16;
  const T* wrow = W + (long long)min(n0 + fr, N - 1) * K;
  const T* xrow = x + (long long)min(fr, B - 1) * a.ldx;  // B operand: 16 streams x 64 B per step, straight from L2
  constexpr int XS = 5;  // activation fragments requested per sub-batch (L2-resident: short latency)
  u32x4_t wq[SK_MAXS], xq[XS];
  auto load_w = [&](int s0) {
#pragma unroll
    for (int i = 0; i < SK_MAXS; ++i) {
      const int st = min(s0 + i, S - 1);  // unconditional (clamped to the row): conditional loads would each get their own wait
      wq[i] = *reinterpret_cast<const u32x4_t*>(wrow + (long long)(st * 4 + kq) * E);
    }
  };
  auto load_x = [&](int s0) {
#pragma unroll
    for (int i = 0; i < XS; ++i) {
      const int st = min(s0 + i, S - 1);
      xq[i] = *reinterpret_cast<const u32x4_t*>(xrow + (long long)(st * 4 + kq) * E);
    }
  };
  // NB on ordering: the vector-memory counter retires loads in order, and a load inside a conditional block gets its own
  // wait.  So every operand is requested unconditionally (clamped addresses, dummy pointers for absent operands), in the
  // order it is needed: epilogue scalars and the (L2-resident) rows for the LayerNorm statistics FIRST, then the HBM
  // weight fragments, then the activation fragments - the statistics are reduced while the weights are in flight.
  const T* bias = reinterpret_cast<const T*>(a.bias);
  const T* res = reinterpret_cast<const T*>(a.res);
  const int ej = tid >> 4, ei = tid & 15;  // epilogue role of threads 0..255: stream ej, tile row ei
  const float* gw_p = has_ln ? a.ln_gw : reinterpret_cast<const float*>(W);
  const float* cb_p = has_ln ? a.ln_cb : reinterpret_cast<const float*>(W);
  const T* bias_p = bias ? bias : W;
  const T* res_p = res ? res : W;
  const DecState* stt_p = a.stt ? a.stt : reinterpret_cast<const DecState*>(W);
  const int cur_pos = stt_p->pos;  // only meaningful when a.stt != null (KV-cache scatter)
  auto load_epi = [&](int n0_, float& e_c, float& e_gw, float& e_res) {
    const int n = min(n0_ + ei, N - 1);
    const float v_gw = gw_p[has_ln ? n : 0];
    const float v_cb = cb_p[has_ln ? n : 0];
    const float v_b = (float)bias_p[bias ? n : 0];
    const float v_r = (float)res_p[res ? (long long)min(ej, B - 1) * a.ldres + n : 0];
    e_gw = v_gw;
    e_c = has_ln ? v_cb : (bias ? v_b : 0.f);
    e_res = res ? v_r : 0.f;
  };
  float e_c0, e_gw0, e_res0;
  load_epi(blockIdx.x * RG * 16, e_c0, e_gw0, e_res0);
  u32x4_t srow[RPW][MAXV];
  if (has_ln) {
#pragma unroll
    for (int rr = 0; rr < RPW; ++rr) {
      const int b = min(wave + NW * rr, B - 1);
#pragma unroll
      for (int i = 0; i < MAXV; ++i) {
        const int vi = min(lane + i * 64, nv_row - 1);
        srow[rr][i] = *reinterpret_cast<const u32x4_t*>(x + (long long)b * a.ldx + vi * E);  // unconditional, clamped
      }
    }
  }
  load_w(s_lo);
  load_x(s_lo);

  if (has_ln) {  // (mean, rstd) of the raw rows: wavefront w owns streams w, w+NW (both in flight together)
    float v[RPW][MAXV][E];
    float sm[RPW], sq[RPW];
#pragma unroll
    for (int rr = 0; rr < RPW; ++rr) {
      sm[rr] = 0.f;
#pragma unroll
      for (int i = 0; i < MAXV; ++i) {
        unpack16<T>(srow[rr][i], v[rr][i]);
        const bool on = (lane + i * 64) < nv_row;
#pragma unroll
        for (int e = 0; e < E; ++e) { v[rr][i][e] = on ? v[rr][i][e] : 0.f; sm[rr] += v[rr][i][e]; }
      }
    }
#pragma unroll
    for (int rr = 0; rr < RPW; ++rr) sm[rr] = wave_sum(sm[rr]) / (float)K;
#pragma unroll
    for (int rr = 0; rr < RPW; ++rr) {
      sq[rr] = 0.f;
#pragma unroll
      for (int i = 0; i < MAXV; ++i) {
        const bool on = (lane + i * 64) < nv_row;
#pragma unroll
        for (int e = 0; e < E; ++e) { const float c = on ? v[rr][i][e] - sm[rr] : 0.f; sq[rr] += c * c; }
      }
    }
#pragma unroll
    for (int rr = 0; rr < RPW; ++rr) {
      const float rstd = 1.0f / sqrtf(wave_sum(sq[rr]) / (float)K + 1e-5f);
      const int b = wave + NW * rr;
      if (lane == 0 && b < B) { stat[b][0] = sm[rr]; stat[b][1] = rstd; }
    }
  }

  for (int grp = 0; grp < RG; ++grp) {
    float e_c = e_c0, e_gw = e_gw0, e_res = e_res0;
    if (grp > 0) {
      n0 = (blockIdx.x * RG + grp) * 16;
      if (n0 >= N) break;
      wrow = W + (long long)min(n0 + fr, N - 1) * K;
      load_epi(n0, e_c, e_gw, e_res);
      load_w(s_lo);
      load_x(s_lo);
    }
    const bool e_on = tid < 256 && (n0 + ei) < N && ej < B;
    f32x4_t acc = f32x4_t{0.f, 0.f, 0.f, 0.f};
    for (int s0 = s_lo; s0 < s_hi; s0 += SK_MAXS) {
      if (s0 > s_lo) { load_w(s0); load_x(s0); }
#pragma unroll
      for (int h = 0; h < SK_MAXS / XS; ++h) {
        if (h > 0) load_x(s0 + h * XS);
#pragma unroll
        for (int i = 0; i < XS; ++i)
          if (s0 + h * XS + i < s_hi) acc = sk_mfma<T>(wq[h * XS + i], xq[i], acc);
      }
    }
    // D[i = weight row (lane>>4)*4 + reg][j = stream lane&15]
    if (grp > 0) __syncthreads();  // previous group's readers are done with `red`
#pragma unroll
    for (int r = 0; r < 4; ++r) red[wave * 256 + (kq * 4 + r) * 16 + fr] = acc[r];
    __syncthreads();
    if (tid < 256) {
      const int j = ej, i = ei;  // stream, row: 16 consecutive rows of one stream per 16 threads
      float v = 0.f;
#pragma unroll
      for (int w = 0; w < NW; ++w) v += red[w * 256 + i * 16 + j];
      const int n = n0 + i;
      if (e_on) {
        if (has_ln) v = stat[j][1] * (v - stat[j][0] * e_gw) + e_c;
        else v += e_c;
        if (a.gelu) v = gelu_exact(v);
        v += e_res;
        if (a.y_f32) {
          a.y_f32[(long long)j * N + n] = v;
        } else if (a.kcache && n >= a.d_model) {
          const int seg = n / a.d_model;
          T* dst = reinterpret_cast<T*>(seg == 1 ? a.kcache : a.vcache);
          dst[(long long)j * a.cache_bstride + (long long)cur_pos * a.d_model + (n - seg * a.d_model)] = (T)v;
        } else {
          reinterpret_cast<T*>(a.y)[(long long)j * a.ldy + n] = (T)v;
        }
      }
    }
  }
}

// W[n,:] *= g (in place, rounded to T); gw[n] = sum_k g[k] W[n,k]; cb[n] = sum_k beta[k] W[n,k] + bias[n]   (one wave per row)
template <typename T>
__global__ __launch_bounds__(256) void fold_ln_kernel(T* __restrict__ W, const T* __restrict__ Wsrc, const T* __restrict__ g,
                                                       const T* __restrict__ beta, const T* __restrict__ bias,
                                                       float* __restrict__ gw, float* __restrict__ cb, int N, int K) {
  const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (n >= N) return;
  float sg = 0.f, sb = 0.f;
  for (int k = lane; k < K; k += 64) {
    const float w = (float)Wsrc[(long long)n * K + k];
    const float gk = (float)g[k];
    sg += gk * w;
    sb += (float)beta[k] * w;
    W[(long long)n * K + k] = (T)(gk * w);
  }
  sg = wave_sum(sg);
  sb = wave_sum(sb);
  if (lane == 0) {
    gw[n] = sg;
    cb[n] = sb + (bias ? (float)bias[n] : 0.f);
  }
}

// ---------------------------------------------------------------------------------------------
// single-query attention (decoder self-attention over the growing cache, cross-attention over the
// cached encoder K/V).  One workgroup per (stream, head); thread = (key group kg, dim quad dq):
// 16 lanes cover one 128-B (bf16) K or V row, NT/16 rows per wave instruction, U rows per thread in
// flight.  Two passes with all loads of a pass issued back to back (the kernel is latency-, not
// bandwidth-bound: ~130 KB per head): scores -> LDS, block max / sum, then P.V.
// ---------------------------------------------------------------------------------------------
template <typename T> __device__ __forceinline__ void load4(const T* p, float* o);
template <> __device__ __forceinline__ void load4<float>(const float* p, float* o) {
  const f32x4_t v = *reinterpret_cast<const f32x4_t*>(p);
  o[0] = v[0]; o[1] = v[1]; o[2] = v[2]; o[3] = v[3];
}
template <> __device__ __forceinline__ void load4<bf16_t>(const bf16_t* p, float* o) {
  typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4_t;
  const bf16x4_t v = *reinterpret_cast<const bf16x4_t*>(p);
  o[0] = (float)v[0]; o[1] = (float)v[1]; o[2] = (float)v[2]; o[3] = (float)v[3];
}

__device__ __forceinline__ float group16_sum(float v) { return row16_sum(v); }
__device__ __forceinline__ float wave_max(float v) { return tw_wave_max(v); }

// Shared by both attention kernels.  NT threads; sc: LDS float[n_keys]; red: LDS float[NT/64 * 64 + 16].
// Returns (in every thread) 1/L; the normalised context vector is written by the first 64 threads.
template <typename T, int NT>
__device__ __forceinline__ float attend_block(const T* __restrict__ qptr, const T* __restrict__ kbase,
                                              const T* __restrict__ vbase, long long stride, int n_keys, float* sc,
                                              float* red, T* __restrict__ outp) {
  constexpr int KG = NT / 16;  // key groups
  constexpr int U = 8;         // keys per thread in flight
  constexpr int NW = NT / 64;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int kg = tid >> 4, dq = tid & 15;
  float qv[4];
  load4<T>(qptr + dq * 4, qv);
  // ---- pass 1: scores ----
  for (int c = 0; c < n_keys; c += KG * U) {
    float kv[U][4];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      int t = c + u * KG + kg;
      if (t >= n_keys) t = n_keys - 1;
      load4<T>(kbase + (long long)t * stride + dq * 4, kv[u]);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      float s = qv[0] * kv[u][0] + qv[1] * kv[u][1] + qv[2] * kv[u][2] + qv[3] * kv[u][3];
      s = group16_sum(s);
      const int t = c + u * KG + kg;
      if (dq == 0 && t < n_keys) sc[t] = s;
    }
  }
  __syncthreads();
  // ---- block max, exp, block sum ----
  float m = -1.0e30f;
  for (int t = tid; t < n_keys; t += NT) m = fmaxf(m, sc[t]);
  m = wave_max(m);
  if (lane == 0) red[wave] = m;
  __syncthreads();
  float M = red[0];
#pragma unroll
  for (int w = 1; w < NW; ++w) M = fmaxf(M, red[w]);
  __syncthreads();
  float ls = 0.f;
  for (int t = tid; t < n_keys; t += NT) {
    const float p = expf(sc[t] - M);
    sc[t] = p;
    ls += p;
  }
  ls = wave_sum(ls);
  if (lane == 0) red[wave] = ls;
  __syncthreads();
  float L = 0.f;
#pragma unroll
  for (int w = 0; w < NW; ++w) L += red[w];
  const float inv = 1.0f / L;
  __syncthreads();
  // ---- pass 2: context = sum_t p[t] * V[t] ----
  float o[4] = {0.f, 0.f, 0.f, 0.f};
  for (int c = 0; c < n_keys; c += KG * U) {
    float vv[U][4];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      int t = c + u * KG + kg;
      if (t >= n_keys) t = n_keys - 1;
      load4<T>(vbase + (long long)t * stride + dq * 4, vv[u]);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int t = c + u * KG + kg;
      const float p = (t < n_keys) ? sc[t] : 0.f;
#pragma unroll
      for (int i = 0; i < 4; ++i) o[i] = fmaf(p, vv[u][i], o[i]);
    }
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {  // fold the 4 key groups of this wavefront
    o[i] = tw_xor32_sum(tw_xor16_sum(o[i]));
  }
  float* wo = red + 16;  // [NW][64]
  if (lane < 16) {
#pragma unroll
    for (int i = 0; i < 4; ++i) wo[wave * 64 + lane * 4 + i] = o[i];
  }
  __syncthreads();
  if (tid < 64) {
    float v = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) v += wo[w * 64 + tid];
    outp[tid] = (T)(v * inv);
  }
  return inv;
}

// Single-round-trip variant for n_keys <= (NT/16)*U: every thread requests its U K rows AND U V rows back to
// back (one memory latency for the whole head), scores and probabilities stay in registers, two barriers.
template <typename T, int NT, int U>
__device__ __forceinline__ float attend_block_fused(const T* __restrict__ qptr, const T* __restrict__ kbase,
                                                    const T* __restrict__ vbase, long long stride, int n_keys,
                                                    int max_rows, float* sc, float* red, T* __restrict__ outp,
                                                    bool want_probs) {
  // `max_rows` (a launch constant) bounds the addresses so that the K/V requests do not depend on `n_keys`, which
  // for the self-attention cache is itself loaded from device memory (DecState.pos): all 2U+1 loads leave at once.
  constexpr int KG = NT / 16;
  constexpr int NW = NT / 64;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int kg = tid >> 4, dq = tid & 15;
  typedef __attribute__((ext_vector_type(4))) float f4;
  float kv[U][4], vv[U][4];
#pragma unroll
  for (int u = 0; u < U; ++u) load4<T>(kbase + (long long)min(u * KG + kg, max_rows - 1) * stride + dq * 4, kv[u]);
#pragma unroll
  for (int u = 0; u < U; ++u) load4<T>(vbase + (long long)min(u * KG + kg, max_rows - 1) * stride + dq * 4, vv[u]);
  float qv[4];
  load4<T>(qptr + dq * 4, qv);
  float sv[U];
  float m = -1.0e30f;
#pragma unroll
  for (int u = 0; u < U; ++u) {
    float s = qv[0] * kv[u][0] + qv[1] * kv[u][1] + qv[2] * kv[u][2] + qv[3] * kv[u][3];
    s = group16_sum(s);
    const bool valid = (u * KG + kg) < n_keys;
    sv[u] = valid ? s : -1.0e30f;
    m = fmaxf(m, sv[u]);
  }
  m = tw_xor32_max(tw_xor16_max(m));  // rows of the wavefront hold different key groups
  if (lane == 0) red[wave] = m;
  __syncthreads();
  float M = red[0];
#pragma unroll
  for (int w = 1; w < NW; ++w) M = fmaxf(M, red[w]);
  float o[4] = {0.f, 0.f, 0.f, 0.f};
  float ls = 0.f;
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const int t = u * KG + kg;
    const float p = (t < n_keys) ? expf(sv[u] - M) : 0.f;
    ls += p;
    if (want_probs && dq == 0 && t < n_keys) sc[t] = p;
#pragma unroll
    for (int i = 0; i < 4; ++i) o[i] = fmaf(p, vv[u][i], o[i]);
  }
  // all 16 lanes of a key group carry the same p: fold the 4 groups of the wavefront, one partial per wave
  ls = tw_xor32_sum(tw_xor16_sum(ls));
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    o[i] = tw_xor32_sum(tw_xor16_sum(o[i]));
  }
  float* wl = red + 8;    // [NW]
  float* wo = red + 16;   // [NW][64]
  if (lane == 0) wl[wave] = ls;
  if (lane < 16) {
#pragma unroll
    for (int i = 0; i < 4; ++i) wo[wave * 64 + lane * 4 + i] = o[i];
  }
  __syncthreads();
  float L = 0.f;
#pragma unroll
  for (int w = 0; w < NW; ++w) L += wl[w];
  const float inv = 1.0f / L;
  if (tid < 64) {
    float v = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) v += wo[w * 64 + tid];
    outp[tid] = (T)(v * inv);
  }
  return inv;
}

template <typename T, bool FUSED>
__global__ __launch_bounds__(256) void dec_self_attn_kernel(const T* __restrict__ q, const T* __restrict__ kc,
                                                             const T* __restrict__ vc, long long cache_bstride,
                                                             T* __restrict__ out, int H, int max_rows,
                                                             const DecState* __restrict__ stt) {
  __shared__ float sc[512];
  __shared__ float red[16 + 4 * 64];
  const int h = blockIdx.x, b = blockIdx.y;
  const int d = H * 64;
  const int n_keys = stt->pos + 1;
  if (FUSED)  // the host guarantees pos < max_rows <= 256 for this call (single round trip, loads do not wait for `pos`)
    attend_block_fused<T, 256, 16>(q + (long long)b * d + h * 64, kc + (long long)b * cache_bstride + h * 64,
                                   vc + (long long)b * cache_bstride + h * 64, d, n_keys, max_rows, sc, red,
                                   out + (long long)b * d + h * 64, false);
  else
    attend_block<T, 256>(q + (long long)b * d + h * 64, kc + (long long)b * cache_bstride + h * 64,
                         vc + (long long)b * cache_bstride + h * 64, d, n_keys, sc, red, out + (long long)b * d + h * 64);
}

template <typename T>
__global__ __launch_bounds__(512) void dec_cross_attn_kernel(const T* __restrict__ q, const T* __restrict__ ck,
                                                              const T* __restrict__ cv, T* __restrict__ out, int H,
                                                              int Tlen, const int* __restrict__ align_slot,
                                                              float* __restrict__ align, int Ha, int P,
                                                              const DecState* __restrict__ stt) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* sc = reinterpret_cast<float*>(smem);  // [Tlen] scores -> unnormalised probabilities
  __shared__ float red[16 + 8 * 64];
  const int h = blockIdx.x, b = blockIdx.y;
  const int d = H * 64;
  const long long base = ((long long)b * H + h) * Tlen * 64;
  const int slot = align_slot ? align_slot[h] : -1;
  float inv;
  if (Tlen <= 32 * 16)
    inv = attend_block_fused<T, 512, 16>(q + (long long)b * d + h * 64, ck + base, cv + base, 64, Tlen, Tlen, sc, red,
                                         out + (long long)b * d + h * 64, slot >= 0);
  else
    inv = attend_block<T, 512>(q + (long long)b * d + h * 64, ck + base, cv + base, 64, Tlen, sc, red,
                               out + (long long)b * d + h * 64);
  if (slot >= 0) __syncthreads();
  if (slot >= 0) {  // A11 side output: softmax row of an alignment head
    float* row = align + (((long long)b * Ha + slot) * P + stt->pos) * Tlen;
    for (int t = threadIdx.x; t < Tlen; t += 512) row[t] = sc[t] * inv;
  }
}

// ---------------------------------------------------------------------------------------------
// sampler: logits processors + argmax
// ---------------------------------------------------------------------------------------------
struct MaxIdx { float v; int i; };
__device__ __forceinline__ MaxIdx better(MaxIdx a, MaxIdx b) {
  // larger value wins; on ties the lower index (torch.argmax returns the first maximal index)
  if (b.v > a.v || (b.v == a.v && b.i < a.i)) return b;
  return a;
}
__device__ __forceinline__ MaxIdx wave_best(MaxIdx x) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    MaxIdx y;
    y.v = __shfl_xor(x.v, o, 64);
    y.i = __shfl_xor(x.i, o, 64);
    x = better(x, y);
  }
  return x;
}

__global__ __launch_bounds__(1024) void sampler_kernel(SamplerArgs a) {
  __shared__ MaxIdx red_text[16], red_ts[16];
  __shared__ float red_sum[16];
  __shared__ int s_choice;
  const int b = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int pos = a.stt->pos;
  const int n_prompt = a.stt->n_prompt;
  const int cur_len = pos + 1;
  int* seq = a.seq + (long long)b * a.seq_ld;
  if (cur_len < n_prompt) {  // still consuming the forced prompt
    if (tid == 0) a.cur_ids[b] = seq[cur_len];
    return;
  }
  if (a.finished[b]) {  // HF: finished rows keep receiving pad_token_id
    if (tid == 0) { seq[cur_len] = a.pad; a.cur_ids[b] = a.pad; }
    return;
  }
  const int V = a.V;
  const bool first = (cur_len == n_prompt);
  const int n_new = cur_len - n_prompt;
  const int ts_begin = a.timestamps ? a.no_ts_id + 1 : V;
  // ---- mask description (uniform scalars) ----
  bool last_ts = false, penult_ts = true;
  int lastts_tok = -1;
  if (a.timestamps) {
    last_ts = (n_new >= 1) && (seq[cur_len - 1] >= ts_begin);
    penult_ts = (n_new < 2) || (seq[cur_len - 2] >= ts_begin);
    lastts_tok = a.last_ts[b];
  }
  int b_lo = 0, b_hi = 0;  // range masked by the pairing rule
  if (last_ts) {
    if (penult_ts) { b_lo = ts_begin; b_hi = V; } else { b_lo = 0; b_hi = a.eos; }
  }
  int c_lo = 0, c_hi = 0;  // non-decreasing timestamps
  if (a.timestamps && lastts_tok >= 0) {
    c_lo = ts_begin;
    c_hi = (last_ts && !penult_ts) ? lastts_tok : lastts_tok + 1;
  }
  int d_hi = 0, e_lo = V;  // first sampled token must be a timestamp <= max_initial
  if (a.timestamps && first) {
    d_hi = ts_begin;
    if (a.max_initial_ts >= 0) e_lo = ts_begin + a.max_initial_ts + 1;
  }
  const bool mask_eos = n_new < a.min_new;
  const float* lg = a.logits + (long long)b * V;
  auto score = [&](int v) -> float {
    bool masked = false;
    if (mask_eos && v == a.eos) masked = true;
    if (a.timestamps && v == a.no_ts_id) masked = true;
    if (v >= b_lo && v < b_hi) masked = true;
    if (v >= c_lo && v < c_hi) masked = true;
    if (v < d_hi) masked = true;
    if (v >= e_lo) masked = true;
    if (first)
      for (int i = 0; i < a.n_begin_suppress; ++i) masked |= (v == a.begin_suppress[i]);
    for (int i = 0; i < a.n_suppress; ++i) masked |= (v == a.suppress[i]);
    return masked ? -INFINITY : lg[v];
  };
  // ---- the row is read ONCE into registers (2 consecutive floats per thread and iteration, all loads in
  //      flight together: the row is 207 KB, the kernel is latency-bound), then both passes run on registers ----
  constexpr int MAXIT = 26;  // 26 * 1024 * 2 = 53248 >= vocab
  const bool vec_ok = (V <= MAXIT * 2048) && ((V & 1) == 0);
  float s0[MAXIT], s1[MAXIT];
  if (vec_ok) {
#pragma unroll
    for (int it = 0; it < MAXIT; ++it) {
      const int v = (it * 1024 + tid) * 2;
      float2 x2 = make_float2(-INFINITY, -INFINITY);
      if (v < V) x2 = *reinterpret_cast<const float2*>(lg + v);
      s0[it] = x2.x;
      s1[it] = x2.y;
    }
  }
  auto masked_at = [&](int v) -> bool {
    bool masked = false;
    if (mask_eos && v == a.eos) masked = true;
    if (a.timestamps && v == a.no_ts_id) masked = true;
    if (v >= b_lo && v < b_hi) masked = true;
    if (v >= c_lo && v < c_hi) masked = true;
    if (v < d_hi) masked = true;
    if (v >= e_lo) masked = true;
    if (first)
      for (int i = 0; i < a.n_begin_suppress; ++i) masked |= (v == a.begin_suppress[i]);
    for (int i = 0; i < a.n_suppress; ++i) masked |= (v == a.suppress[i]);
    return masked;
  };
  // ---- pass 1: best text token, best timestamp token ----
  MaxIdx bt{-INFINITY, 0x7fffffff}, bs{-INFINITY, 0x7fffffff};
  if (vec_ok) {
#pragma unroll
    for (int it = 0; it < MAXIT; ++it) {
      const int v = (it * 1024 + tid) * 2;
      if (v < V) {
        if (masked_at(v)) s0[it] = -INFINITY;
        if (masked_at(v + 1)) s1[it] = -INFINITY;
        MaxIdx c0{s0[it], v}, c1{s1[it], v + 1};
        if (v < ts_begin) bt = better(bt, c0); else bs = better(bs, c0);
        if (v + 1 < ts_begin) bt = better(bt, c1); else bs = better(bs, c1);
      }
    }
  } else {
    for (int v = tid; v < V; v += 1024) {
      const float sc_ = score(v);
      MaxIdx c{sc_, v};
      if (v < ts_begin) bt = better(bt, c); else bs = better(bs, c);
    }
  }
  bt = wave_best(bt);
  bs = wave_best(bs);
  if (lane == 0) { red_text[wave] = bt; red_ts[wave] = bs; }
  __syncthreads();
  bt = red_text[0];
  bs = red_ts[0];
  for (int w = 1; w < 16; ++w) { bt = better(bt, red_text[w]); bs = better(bs, red_ts[w]); }
  // ---- pass 2: logsumexp over timestamp tokens ----
  bool force_ts = false;
  if (a.timestamps && bs.v > -INFINITY) {
    float sum = 0.f;
    if (vec_ok) {
#pragma unroll
      for (int it = 0; it < MAXIT; ++it) {
        const int v = (it * 1024 + tid) * 2;
        if (v < V) {
          if (v >= ts_begin) sum += expf(s0[it] - bs.v);
          if (v + 1 >= ts_begin) sum += expf(s1[it] - bs.v);
        }
      }
    } else {
      for (int v = ts_begin + tid; v < V; v += 1024) sum += expf(score(v) - bs.v);
    }
    sum = wave_sum(sum);
    if (lane == 0) red_sum[wave] = sum;
    __syncthreads();
    float tot = 0.f;
    for (int w = 0; w < 16; ++w) tot += red_sum[w];
    const float lse_ts = bs.v + logf(tot);
    force_ts = lse_ts > bt.v;
  }
  if (tid == 0) {
    int choice;
    if (force_ts) choice = bs.i;
    else choice = (bs.v > bt.v) ? bs.i : bt.i;
    if (choice == 0x7fffffff) choice = 0;  // everything masked: torch.argmax of all -inf is 0
    seq[cur_len] = choice;
    a.cur_ids[b] = choice;
    if (a.timestamps && choice >= ts_begin) a.last_ts[b] = choice;
    if (choice == a.eos) a.finished[b] = 1;
    s_choice = choice;
  }
}

__global__ void advance_kernel(DecState* stt) { stt->pos += 1; }

}  // namespace

template <typename T, int BT>
static hipError_t gemv_r(const GemvArgs& a, hipStream_t st) {
  constexpr int E = ElemTraits<T>::kPer16B;
  constexpr int CV = GemvChunk<T, BT>::kVec;
  if (a.K % E != 0 || (a.ln_gw && a.K > 1280) || a.B > BT) return hipErrorInvalidValue;
  const int nv_row = a.K / E;
  const bool chunked = nv_row > CV;
  if (chunked && a.ln_gw) return hipErrorInvalidValue;
  const size_t lds = (size_t)BT * (chunked ? CV : nv_row) * 16;
  // rows per wavefront: keep >= ~1000 wavefronts in flight when N allows it; very tall matrices (the tied
  // logits projection) additionally walk GEMV_RG row groups per workgroup so x is staged/normalised once per 128 rows
  if (a.N >= 16384) {
    const int rows_per_block = 4 * 4 * GEMV_RG;
    dim3 grid((a.N + rows_per_block - 1) / rows_per_block);
    hipLaunchKernelGGL((gemv_kernel<T, BT, 4, true>), grid, dim3(256), lds, st, a);
  } else if (a.N >= 3072) {
    dim3 grid((a.N + 7) / 8);
    hipLaunchKernelGGL((gemv_kernel<T, BT, 2, false>), grid, dim3(256), lds, st, a);
  } else {
    dim3 grid((a.N + 3) / 4);
    hipLaunchKernelGGL((gemv_kernel<T, BT, 1, false>), grid, dim3(256), lds, st, a);
  }
  return hipGetLastError();
}

template <typename T, int BT>
static hipError_t gemv_attr() {
  // up to 80 KiB of staged activations: raise the dynamic-LDS cap once per instantiation
  const int cap = GEMV_LDS_BUDGET;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemv_kernel<T, BT, 1, false>), hipFuncAttributeMaxDynamicSharedMemorySize, cap);
  if (e != hipSuccess) return e;
  e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemv_kernel<T, BT, 2, false>), hipFuncAttributeMaxDynamicSharedMemorySize, cap);
  if (e != hipSuccess) return e;
  return hipFuncSetAttribute(reinterpret_cast<const void*>(&gemv_kernel<T, BT, 4, true>), hipFuncAttributeMaxDynamicSharedMemorySize, cap);
}

hipError_t init_decode_kernels() {
  hipError_t e;
  if ((e = gemv_attr<bf16_t, 1>()) != hipSuccess) return e;
  if ((e = gemv_attr<bf16_t, 4>()) != hipSuccess) return e;
  if ((e = gemv_attr<bf16_t, 8>()) != hipSuccess) return e;
  if ((e = gemv_attr<bf16_t, 16>()) != hipSuccess) return e;
  if ((e = gemv_attr<float, 1>()) != hipSuccess) return e;
  if ((e = gemv_attr<float, 4>()) != hipSuccess) return e;
  if ((e = gemv_attr<float, 8>()) != hipSuccess) return e;
  return gemv_attr<float, 16>();
}

static int env_int(const char* name, int dflt) {
  const char* e = getenv(name);
  return e ? atoi(e) : dflt;
}

template <typename T, int NW>
static hipError_t skinny_launch_nw(const GemvArgs& a0, hipStream_t st) {
  constexpr int E = ElemTraits<T>::kPer16B;
  GemvArgs a = a0;
  if (a.K % (4 * E) != 0 || (a.ln_gw && a.K > 1280) || a.B > 16) return hipErrorInvalidValue;
  const size_t lds = (size_t)NW * 256 * 4;
  const int tiles = (a.N + 15) / 16;
  // at most `max_blocks` workgroups: tall matrices (the tied logits projection) walk several tiles per workgroup
  static const int max_blocks = env_int("TW_SK_MAX_BLOCKS", 512);
  a.rg = (tiles + max_blocks - 1) / max_blocks;
  if (a.rg < 1) a.rg = 1;
  dim3 grid((tiles + a.rg - 1) / a.rg);
  const int steps_per_wave = (a.K / E / 4 + NW - 1) / NW;
  if (steps_per_wave <= 5)
    hipLaunchKernelGGL((skinny_mfma_kernel<T, NW, 5>), grid, dim3(NW * 64), lds, st, a);
  else
    hipLaunchKernelGGL((skinny_mfma_kernel<T, NW, 10>), grid, dim3(NW * 64), lds, st, a);
  return hipGetLastError();
}

template <typename T>
static hipError_t skinny_launch(const GemvArgs& a, hipStream_t st) {
  static const int nw_big = env_int("TW_SK_NW_BIGK", 8);  // wavefronts per tile when K is long (fc2: K = 5120)
  if (a.K >= 4096 && nw_big == 16) return skinny_launch_nw<T, 16>(a, st);
  return skinny_launch_nw<T, 8>(a, st);
}

static int gemv_mfma_min_b() {  // streams from which the MFMA formulation is used (TW_SKINNY_MIN_B overrides, for A/B runs)
  static int v = -1;
  if (v < 0) {
    const char* e = getenv("TW_SKINNY_MIN_B");
    v = e ? atoi(e) : 5;
    if (v < 1) v = 1;
  }
  return v;
}

template <typename T>
static hipError_t gemv_b(const GemvArgs& a, hipStream_t st) {
  if (a.B >= gemv_mfma_min_b() && a.B <= 16) return skinny_launch<T>(a, st);
  if (a.B <= 1) return gemv_r<T, 1>(a, st);
  if (a.B <= 4) return gemv_r<T, 4>(a, st);
  if (a.B <= 8) return gemv_r<T, 8>(a, st);
  if (a.B <= 16) return gemv_r<T, 16>(a, st);
  return hipErrorInvalidValue;
}

hipError_t launch_gemv(int dtype, const GemvArgs& a, hipStream_t st) {
  return dtype == 1 ? gemv_b<bf16_t>(a, st) : gemv_b<float>(a, st);
}

hipError_t launch_dec_self_attn(int dtype, const void* q, const void* kc, const void* vc, long long cache_bstride,
                                void* out, int B, int H, int key_bound, const DecState* stt, hipStream_t st) {
  // key_bound: an upper bound of pos+1 for this call known on the host (prompt + max new tokens), also <= cache rows
  const bool fused = key_bound <= 256;
  if (dtype == 1) {
    if (fused)
      hipLaunchKernelGGL((dec_self_attn_kernel<bf16_t, true>), dim3(H, B), dim3(256), 0, st, (const bf16_t*)q,
                         (const bf16_t*)kc, (const bf16_t*)vc, cache_bstride, (bf16_t*)out, H, key_bound, stt);
    else
      hipLaunchKernelGGL((dec_self_attn_kernel<bf16_t, false>), dim3(H, B), dim3(256), 0, st, (const bf16_t*)q,
                         (const bf16_t*)kc, (const bf16_t*)vc, cache_bstride, (bf16_t*)out, H, key_bound, stt);
  } else {
    if (fused)
      hipLaunchKernelGGL((dec_self_attn_kernel<float, true>), dim3(H, B), dim3(256), 0, st, (const float*)q,
                         (const float*)kc, (const float*)vc, cache_bstride, (float*)out, H, key_bound, stt);
    else
      hipLaunchKernelGGL((dec_self_attn_kernel<float, false>), dim3(H, B), dim3(256), 0, st, (const float*)q,
                         (const float*)kc, (const float*)vc, cache_bstride, (float*)out, H, key_bound, stt);
  }
  return hipGetLastError();
}

hipError_t launch_dec_cross_attn(int dtype, const void* q, const void* ck, const void* cv, void* out, int B, int H,
                                 int T, const int* align_slot_for_head, float* align, int Ha, int P,
                                 const DecState* stt, hipStream_t st) {
  const size_t lds = (size_t)T * sizeof(float);
  if (dtype == 1)
    hipLaunchKernelGGL(dec_cross_attn_kernel<bf16_t>, dim3(H, B), dim3(512), lds, st, (const bf16_t*)q,
                       (const bf16_t*)ck, (const bf16_t*)cv, (bf16_t*)out, H, T, align_slot_for_head, align, Ha, P, stt);
  else
    hipLaunchKernelGGL(dec_cross_attn_kernel<float>, dim3(H, B), dim3(512), lds, st, (const float*)q, (const float*)ck,
                       (const float*)cv, (float*)out, H, T, align_slot_for_head, align, Ha, P, stt);
  return hipGetLastError();
}

hipError_t launch_sampler(const SamplerArgs& a, hipStream_t st) {
  hipLaunchKernelGGL(sampler_kernel, dim3(a.B), dim3(1024), 0, st, a);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(advance_kernel, dim3(1), dim3(1), 0, st, a.stt);
  return hipGetLastError();
}

hipError_t launch_advance(DecState* stt, hipStream_t st) {
  hipLaunchKernelGGL(advance_kernel, dim3(1), dim3(1), 0, st, stt);
  return hipGetLastError();
}

hipError_t launch_fold_ln(int dtype, void* W, const void* Wsrc, const void* g, const void* beta, const void* bias, float* gw,
                          float* cb, int N, int K, hipStream_t st) {
  dim3 grid((N + 3) / 4);
  if (dtype == 1)
    hipLaunchKernelGGL(fold_ln_kernel<bf16_t>, grid, dim3(256), 0, st, (bf16_t*)W, (const bf16_t*)Wsrc, (const bf16_t*)g,
                       (const bf16_t*)beta, (const bf16_t*)bias, gw, cb, N, K);
  else
    hipLaunchKernelGGL(fold_ln_kernel<float>, grid, dim3(256), 0, st, (float*)W, (const float*)Wsrc, (const float*)g,
                       (const float*)beta, (const float*)bias, gw, cb, N, K);
  return hipGetLastError();
}
