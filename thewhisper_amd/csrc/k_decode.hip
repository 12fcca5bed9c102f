// A6-A8, A10: the per-token decoder path.  With one new token per stream every projection is a
// skinny GEMM  y[B<=16, N] = f(x[B, K]) . W[N, K]^T  whose cost is streaming W once from HBM
// (1.6 GB per step for large-v3 in bf16), so these kernels are HBM-bound by construction:
//
//  * skinny_mfma_kernel (launch_gemv): the <=16 activation rows are exactly one MFMA operand tile, so one 16-row weight
//    tile per workgroup is streamed ONCE (fragment-major layout prepared at load time: every wavefront request is 1 KiB
//    contiguous) and contracted by v_mfma_f32_16x16x32_bf16 (f32 mode: 16x16x4_f32) for all streams at once; K is split
//    over the wavefronts and summed in a fixed order through LDS (deterministic, no atomics).  Fused: folded
//    pre-LayerNorm, bias, GELU, residual, KV-cache scatter, fp32 logits.
//  * dec_self_attn / dec_cross_attn: single-query attention over the KV cache, one wavefront per
//    (stream, head) [x4 for the 500..1500-key cross attention], lanes = 4 key groups x 16 dim
//    quads, online softmax in registers, 16-lane shuffle dot products.  Cross attention also emits
//    the softmax rows of the alignment heads for the DTW (A11).
//  * sampler: Whisper's logits processors + first-index argmax on the fp32 logits, one workgroup
//    per stream (HF:generation/logits_process.py:1816-2047, HF:generation/utils.py:2925).
#include "tw_common.h"

#include <cstdlib>

namespace {

// Probe builds (tools/dbg/probe_gemv.hip -DTW_PROBE_TS) stamp s_memrealtime at a few points of the projection kernels.
#ifdef TW_PROBE_TS
__device__ unsigned long long* g_probe_ts;
#define TW_TS(k) do { if (threadIdx.x == 0) g_probe_ts[((size_t)cur_pos * 1024 + blockIdx.x) * 8 + (k)] = wall_clock64(); } while (0)
#else
#define TW_TS(k) do { } while (0)
#endif

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

// Folded pre-LayerNorm.  For y = LN(x) . W^T + bias with LN(x) = (x - mean) * rstd * g + beta:
//     y[n] = rstd * ( x . W'[n]  -  mean * gW[n] )  +  cb[n],
//     W'[n,k] = g[k] W[n,k],   gW[n] = sum_k g[k] W[n,k],   cb[n] = sum_k beta[k] W[n,k] + bias[n]
// W', gW and cb are prepared once at weight-finalisation time (fold_ln_kernel below), so the projection streams W'
// against the RAW residual stream exactly like a projection without LayerNorm, and only the two per-stream scalars
// (mean, rstd) have to be known when the epilogue runs - they are computed off the critical path while the weight
// loads are in flight.  No normalised copy of x is ever materialised.


// ---------------------------------------------------------------------------------------------
// Skinny MFMA GEMM for 1..16 concurrent streams:  y[b, n0..n0+15] for one 16-row weight tile per
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

// Epilogue flavours (compile-time: a runtime-uniform branch around a load makes hipcc wait for EVERY outstanding load at
// the join, which serialises the whole prologue - see the ordering note below).
enum : int { SK_STORE = 0, SK_RES = 1, SK_GELU = 2, SK_KV = 3, SK_F32 = 4 };

template <typename T>
__device__ __forceinline__ u32x4_t sk_load_w(const T* p) {  // weights are read exactly once per step: non-temporal
  return __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(p));
}

// (sum, sum of squares) of the 16 bytes of one activation fragment, accumulated per lane
template <typename T> __device__ __forceinline__ void sk_stats(const u32x4_t& v, float& s, float& ss);
template <> __device__ __forceinline__ void sk_stats<bf16_t>(const u32x4_t& v, float& s, float& ss) {
  const bf16x8_t a = __builtin_bit_cast(bf16x8_t, v);
  const bf16x2_t one = {(bf16_t)1.0f, (bf16_t)1.0f};
  // named pairs, not a loop over v[i]: see dot16 above
  const bf16x2_t p0 = __builtin_shufflevector(a, a, 0, 1), p1 = __builtin_shufflevector(a, a, 2, 3);
  const bf16x2_t p2 = __builtin_shufflevector(a, a, 4, 5), p3 = __builtin_shufflevector(a, a, 6, 7);
  s = __builtin_amdgcn_fdot2_f32_bf16(p0, one, s, false);
  ss = __builtin_amdgcn_fdot2_f32_bf16(p0, p0, ss, false);
  s = __builtin_amdgcn_fdot2_f32_bf16(p1, one, s, false);
  ss = __builtin_amdgcn_fdot2_f32_bf16(p1, p1, ss, false);
  s = __builtin_amdgcn_fdot2_f32_bf16(p2, one, s, false);
  ss = __builtin_amdgcn_fdot2_f32_bf16(p2, p2, ss, false);
  s = __builtin_amdgcn_fdot2_f32_bf16(p3, one, s, false);
  ss = __builtin_amdgcn_fdot2_f32_bf16(p3, p3, ss, false);
}
template <> __device__ __forceinline__ void sk_stats<float>(const u32x4_t& v, float& s, float& ss) {
  const f32x4_t a = __builtin_bit_cast(f32x4_t, v);
#pragma unroll
  for (int i = 0; i < 4; ++i) { s += a[i]; ss = fmaf(a[i], a[i], ss); }
}

// SK_MAXS = weight/activation fragments a wavefront keeps in flight (5 covers K = 1280 in bf16 with 8 wavefronts)
// LN      = folded pre-LayerNorm (ln_gw / ln_cb given), EPI = epilogue flavour, MULTI = several 16-row tiles per
//           workgroup (tall matrices: the tied logits projection), next tile prefetched behind the current reduction.
// Activations (x, the residual operand, and the outputs of SK_RES / SK_GELU) live in the fragment-major "xt" layout of
// tw_common.h (tw_xt_index): like the weights, every wavefront request is then 1 KiB contiguous.  The LayerNorm
// statistics come from the activation fragments the wavefront holds anyway (per-lane sum / sum of squares with
// v_dot2, folded over the 4 k-groups by two lane swaps, over the wavefronts in the epilogue through LDS): no extra
// loads and ~60 instead of ~500 VALU instructions on the critical path (at 4 cycles per wave64 VALU op that was 1 us).
// Ordering rules this kernel follows (measured with tools/dbg/probe_gemv.hip, stamps of s_memrealtime):
//  * all kernel arguments are pinned in SGPRs by one asm statement: one batch of scalar loads, one wait;
//  * vmcnt retires loads IN ORDER, so operands are requested in the order they are consumed: the (L2-resident)
//    activation fragments, then the HBM weight fragments; sched_barriers keep hipcc from reordering the groups or
//    hoisting arithmetic between them;
//  * no load sits inside a conditional block (clamped addresses, zero-selected operands instead): at the join of a
//    runtime-uniform branch around a load hipcc waits for EVERY outstanding load.
template <typename T, int NW, int SK_MAXS, bool LN, int EPI, bool MULTI>
__global__ __launch_bounds__(NW * 64, (NW >= 16 ? 4 : (SK_MAXS <= 5 ? 4 : 2))) void skinny_mfma_kernel(GemvArgs a) {
  constexpr int E = ElemTraits<T>::kPer16B;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  __shared__ float pstat[NW][16][2];  // per wavefront and stream: (sum, sum of squares) over the wavefront's K slice
  const T* x = reinterpret_cast<const T*>(a.x);
  const T* W = reinterpret_cast<const T*>(a.W);
  const T* bias = reinterpret_cast<const T*>(a.bias);
  const T* res = reinterpret_cast<const T*>(a.res);
  const float* gw_p = a.ln_gw;
  const float* cb_p = a.ln_cb;
  const int K = a.K, N = a.N, B = a.B, ldy = a.ldy, RG = a.rg, d_model = a.d_model;
  const long long cache_bstride = a.cache_bstride;
  T* y = reinterpret_cast<T*>(a.y);
  float* y_f32 = a.y_f32;
  T* kcache = reinterpret_cast<T*>(a.kcache);
  T* vcache = reinterpret_cast<T*>(a.vcache);
  const DecState* stt = a.stt;
  asm volatile("" ::"s"(x), "s"(W), "s"(bias), "s"(res), "s"(gw_p), "s"(cb_p), "s"(K), "s"(N), "s"(B), "s"(a.gelu), "s"(ldy),
               "s"(RG), "s"(d_model), "s"(cache_bstride), "s"(y), "s"(y_f32), "s"(kcache), "s"(vcache), "s"(stt));
  int cur_pos = 0;
#ifdef TW_PROBE_TS
  cur_pos = stt->pos;
#else
  if (EPI == SK_KV) cur_pos = stt->pos;
#endif
  TW_TS(0);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 15, kq = lane >> 4;
  const int S = K / (4 * E);  // 64-B steps per row
  const int s_lo = (int)((long long)wave * S / NW), s_hi = (int)((long long)(wave + 1) * S / NW);
  float* red = reinterpret_cast<float*>(smem);  // [NW][256]
  const int n_tiles = (N + 15) / 16;
  const int tile0 = blockIdx.x * RG;
  const int ej = (tid >> 4) & 15, ei = tid & 15;  // epilogue role of threads 0..255: stream ej, tile row ei

  u32x4_t wq[SK_MAXS], xq[SK_MAXS];
  float e_c = 0.f, e_gw = 0.f, e_res = 0.f;
  // --- request helpers: all unconditional, addresses clamped into the matrix ---
  auto load_epi = [&](int tile, float& c, float& g, float& r) {
    const int n = min(tile * 16 + ei, N - 1);
    if (LN) {
      g = gw_p[n];
      c = cb_p[n];
    } else {
      const float v = (float)(bias ? bias : W)[bias ? n : 0];
      c = bias ? v : 0.f;
    }
    if (EPI == SK_RES) r = (float)res[tw_xt_index<T>(ej, n)];
  };
  auto load_w = [&](int tile, int s0) {  // fragment-major weights: 1 KiB contiguous per wavefront request
    const T* wt = W + ((long long)min(tile, n_tiles - 1) * S * 64 + lane) * E;
#pragma unroll
    for (int i = 0; i < SK_MAXS; ++i) wq[i] = sk_load_w<T>(wt + (long long)min(s0 + i, S - 1) * (64 * E));
  };
  auto load_x = [&](int s0) {            // fragment-major activations: lane (stream fr, k-group kq) of step s
    const T* xt = x + (long long)lane * E;
#pragma unroll
    for (int i = 0; i < SK_MAXS; ++i) xq[i] = *reinterpret_cast<const u32x4_t*>(xt + (long long)min(s0 + i, S - 1) * (64 * E));
  };

  load_x(s_lo);
  __builtin_amdgcn_sched_barrier(0);  // request order = consumption order: hipcc must not reorder the groups
  load_w(tile0, s_lo);
  load_epi(tile0, e_c, e_gw, e_res);
  __builtin_amdgcn_sched_barrier(0);  // ... nor hoist arithmetic between the requests
  TW_TS(1);

  float mean = 0.f, rstd = 1.f;
  const int n_grp = MULTI ? RG : 1;
  for (int grp = 0; grp < n_grp; ++grp) {
    const int tile = tile0 + grp;
    f32x4_t acc = f32x4_t{0.f, 0.f, 0.f, 0.f};
    float ps = 0.f, pss = 0.f;
    auto mfma_round = [&](int s0) {
#pragma unroll
      for (int i = 0; i < SK_MAXS; ++i) {
        const bool on = s0 + i < s_hi;  // wave-uniform: steps past this wavefront's K slice contribute zero
        const u32x4_t zero = u32x4_t{0u, 0u, 0u, 0u};
        const u32x4_t xv = on ? xq[i] : zero;
        if (LN && (!MULTI || grp == 0)) sk_stats<T>(xv, ps, pss);
        acc = sk_mfma<T>(on ? wq[i] : zero, xv, acc);
      }
    };
    mfma_round(s_lo);  // operands already in flight
    for (int s0 = s_lo + SK_MAXS; s0 < s_hi; s0 += SK_MAXS) {  // K longer than one round of fragments
      load_x(s0);
      load_w(tile, s0);
      mfma_round(s0);
    }
    // operands of the tile after this one are requested now, behind the reduction and the epilogue of the current one
    // (MULTI is only launched with a single round of fragments per tile, so the activation fragments stay in registers)
    float n_c = 0.f, n_gw = 0.f, n_res = 0.f;
    if (MULTI) {
      const int nt = min(tile + 1, n_tiles - 1);
      load_w(nt, s_lo);
      load_epi(nt, n_c, n_gw, n_res);
    }
    TW_TS(2);
    // D[i = weight row (lane>>4)*4 + reg][j = stream lane&15]
    if (MULTI && grp > 0) __syncthreads();  // previous tile's readers are done with `red`
#pragma unroll
    for (int r = 0; r < 4; ++r) red[wave * 256 + (kq * 4 + r) * 16 + fr] = acc[r];
    if (LN && (!MULTI || grp == 0)) {
      ps = tw_xor32_sum(tw_xor16_sum(ps));
      pss = tw_xor32_sum(tw_xor16_sum(pss));
      if (lane < 16) { pstat[wave][fr][0] = ps; pstat[wave][fr][1] = pss; }
    }
    __syncthreads();
    TW_TS(3);
    if (tid < 256) {
      const int j = ej, i = ei;  // stream, row: 16 consecutive rows of one stream per 16 threads
      float v = 0.f;
#pragma unroll
      for (int w = 0; w < NW; ++w) v += red[w * 256 + i * 16 + j];
      const int n = tile * 16 + i;
      if (LN) {
        if (!MULTI || grp == 0) {
          float sx = 0.f, sxx = 0.f;
#pragma unroll
          for (int w = 0; w < NW; ++w) { sx += pstat[w][j][0]; sxx += pstat[w][j][1]; }
          const float inv_k = __builtin_amdgcn_rcpf((float)K);
          mean = sx * inv_k;
          rstd = __frsqrt_rn(fmaxf(sxx * inv_k - mean * mean, 0.f) + 1e-5f);
        }
        v = rstd * (v - mean * e_gw) + e_c;
      } else {
        v += e_c;
      }
      if (EPI == SK_GELU) v = gelu_exact(v);
      if (EPI == SK_RES) v += e_res;
      if (tile < n_tiles && n < N && j < B) {
        if (EPI == SK_F32) {
          y_f32[(long long)j * N + n] = v;
        } else if (EPI == SK_KV) {
          const int seg = n / d_model;  // 0: query -> y, 1: key -> cache, 2: value -> cache   (one predicated store)
          T* dst = seg == 0 ? y + (long long)j * ldy + n
                            : (seg == 1 ? kcache : vcache) + (long long)j * cache_bstride + (long long)cur_pos * d_model + (n - seg * d_model);
          *dst = (T)v;
        } else if (EPI == SK_STORE) {
          y[(long long)j * ldy + n] = (T)v;  // row-major [B][ldy] (the attention kernels' query operand)
        } else {
          y[tw_xt_index<T>(j, n)] = (T)v;    // feeds the next projection: fragment-major
        }
      }
    }
    if (MULTI) { e_c = n_c; e_gw = n_gw; e_res = n_res; }
    TW_TS(4);
  }
}

// Row-major W[N][K] -> fragment-major layout read by skinny_mfma_kernel: for tile t = n/16 and step s = k/(4E) one
// 1-KiB block holds the MFMA A operand exactly as the 64 lanes consume it, lane l = kq*16 + fr <- row t*16+fr,
// 16-B vector s*4+kq.  A wavefront's K slice is then ONE contiguous run (its requests are full cache lines in address
// order, like a plain streaming copy) instead of 16 row segments of 64 B per request.  Rows >= N are zero.
template <typename T>
__global__ __launch_bounds__(256) void tile_weights_kernel(const T* __restrict__ src, T* __restrict__ dst, int N, int K) {
  constexpr int E = ElemTraits<T>::kPer16B;
  const int S = K / E / 4;
  const long long v = (long long)blockIdx.x * 256 + threadIdx.x;  // destination vector index
  const long long total = (long long)((N + 15) / 16) * S * 64;
  if (v >= total) return;
  const int l = (int)(v & 63);
  const long long ts = v >> 6;
  const int s = (int)(ts % S);
  const int t = (int)(ts / S);
  const int n = t * 16 + (l & 15), kv = s * 4 + (l >> 4);
  u32x4_t val = u32x4_t{0u, 0u, 0u, 0u};
  if (n < N) val = *reinterpret_cast<const u32x4_t*>(src + (long long)n * K + (long long)kv * E);
  *reinterpret_cast<u32x4_t*>(dst + v * E) = val;
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
                                              float* red, T* __restrict__ out_base, int out_j, int out_k0) {
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
    out_base[tw_xt_index<T>(out_j, out_k0 + tid)] = (T)(v * inv);  // feeds o-proj: fragment-major (tw_common.h)
  }
  return inv;
}

// Single-round-trip variant for n_keys <= (NT/16)*U: every thread requests its U K rows AND U V rows back to
// back (one memory latency for the whole head), scores and probabilities stay in registers, two barriers.
template <typename T, int NT, int U>
__device__ __forceinline__ float attend_block_fused(const T* __restrict__ qptr, const T* __restrict__ kbase,
                                                    const T* __restrict__ vbase, long long stride, int n_keys,
                                                    int max_rows, float* sc, float* red, T* __restrict__ out_base,
                                                    int out_j, int out_k0, bool want_probs) {
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
    out_base[tw_xt_index<T>(out_j, out_k0 + tid)] = (T)(v * inv);  // feeds o-proj: fragment-major (tw_common.h)
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
                                   vc + (long long)b * cache_bstride + h * 64, d, n_keys, max_rows, sc, red, out, b, h * 64,
                                   false);
  else
    attend_block<T, 256>(q + (long long)b * d + h * 64, kc + (long long)b * cache_bstride + h * 64,
                         vc + (long long)b * cache_bstride + h * 64, d, n_keys, sc, red, out, b, h * 64);
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
                                         out, b, h * 64, slot >= 0);
  else
    inv = attend_block<T, 512>(q + (long long)b * d + h * 64, ck + base, cv + base, 64, Tlen, sc, red, out, b, h * 64);
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

hipError_t init_decode_kernels() { return hipSuccess; }  // nothing to configure (kept for the call site in tw_create)

static int env_int(const char* name, int dflt) {
  const char* e = getenv(name);
  return e ? atoi(e) : dflt;
}

template <typename T, int NW, int SK_MAXS, bool MULTI>
static hipError_t skinny_launch_v(const GemvArgs& a, dim3 grid, size_t lds, hipStream_t st) {
  const bool ln = a.ln_gw != nullptr;
#define SK_GO(LNV, EPIV) hipLaunchKernelGGL((skinny_mfma_kernel<T, NW, SK_MAXS, LNV, EPIV, MULTI>), grid, dim3(NW * 64), lds, st, a)
  if (a.y_f32) {
    if (!ln || a.res || a.gelu || a.kcache) return hipErrorInvalidValue;
    SK_GO(true, SK_F32);
  } else if (a.kcache) {
    if (!ln || a.res || a.gelu) return hipErrorInvalidValue;
    SK_GO(true, SK_KV);
  } else if (a.gelu) {
    if (!ln || a.res) return hipErrorInvalidValue;
    SK_GO(true, SK_GELU);
  } else if (a.res) {
    if (ln) return hipErrorInvalidValue;
    SK_GO(false, SK_RES);
  } else if (ln) {
    SK_GO(true, SK_STORE);
  } else {
    SK_GO(false, SK_STORE);
  }
#undef SK_GO
  return hipGetLastError();
}

template <typename T, int NW>
static hipError_t skinny_launch_nw(const GemvArgs& a0, hipStream_t st) {
  constexpr int E = ElemTraits<T>::kPer16B;
  GemvArgs a = a0;
  if (a.K % (4 * E) != 0 || a.B > 16) return hipErrorInvalidValue;
  const size_t lds = (size_t)NW * 256 * 4;
  const int tiles = (a.N + 15) / 16;
  // at most `max_blocks` workgroups: tall matrices (the tied logits projection) walk several tiles per workgroup
  static const int max_blocks = env_int("TW_SK_MAX_BLOCKS", 512);
  const int steps_per_wave = (a.K / E / 4 + NW - 1) / NW;
  a.rg = (tiles + max_blocks - 1) / max_blocks;
  if (a.rg < 1 || steps_per_wave > 10) a.rg = 1;  // several tiles per workgroup only with one round of fragments per tile
  dim3 grid((tiles + a.rg - 1) / a.rg);
  if (a.rg > 1) {
    if (steps_per_wave <= 5) return skinny_launch_v<T, NW, 5, true>(a, grid, lds, st);
    return skinny_launch_v<T, NW, 10, true>(a, grid, lds, st);
  }
  if (steps_per_wave <= 5) return skinny_launch_v<T, NW, 5, false>(a, grid, lds, st);
  return skinny_launch_v<T, NW, 10, false>(a, grid, lds, st);
}

template <typename T>
static hipError_t skinny_launch(const GemvArgs& a, hipStream_t st) {
  static const int nw_big = env_int("TW_SK_NW_BIGK", 16);  // wavefronts per tile when K is long (fc2: K = 5120)
  if (a.K >= 4096 && nw_big == 16 && !a.ln_gw && !a.y_f32 && !a.kcache && !a.gelu) return skinny_launch_nw<T, 16>(a, st);
  return skinny_launch_nw<T, 8>(a, st);
}

template <typename T>
static hipError_t gemv_b(const GemvArgs& a, hipStream_t st) {
  if (a.B < 1 || a.B > 16) return hipErrorInvalidValue;
  return skinny_launch<T>(a, st);
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

hipError_t launch_tile_weights(int dtype, const void* src, void* dst, int N, int K, hipStream_t st) {
  const int E = dtype == 1 ? 8 : 4;
  if (K % (4 * E) != 0) return hipErrorInvalidValue;
  const long long total = (long long)((N + 15) / 16) * (K / E / 4) * 64;
  dim3 grid((unsigned)((total + 255) / 256));
  if (dtype == 1) hipLaunchKernelGGL(tile_weights_kernel<bf16_t>, grid, dim3(256), 0, st, (const bf16_t*)src, (bf16_t*)dst, N, K);
  else hipLaunchKernelGGL(tile_weights_kernel<float>, grid, dim3(256), 0, st, (const float*)src, (float*)dst, N, K);
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
