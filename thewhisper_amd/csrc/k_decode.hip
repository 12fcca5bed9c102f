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

namespace {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float gelu_exact(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }

template <typename T> __device__ __forceinline__ float dot16(const u32x4_t& w, const u32x4_t& x, float acc);
template <> __device__ __forceinline__ float dot16<bf16_t>(const u32x4_t& w, const u32x4_t& x, float acc) {
#pragma unroll
  for (int i = 0; i < 4; ++i)
    acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, w[i]), __builtin_bit_cast(bf16x2_t, x[i]), acc,
                                          false);
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

constexpr int GEMV_KC = 1280;  // K chunk staged in LDS (elements)

template <typename T, int BT, int R>
__global__ __launch_bounds__(256) void gemv_kernel(GemvArgs a) {
  constexpr int E = ElemTraits<T>::kPer16B;
  constexpr int MAXV = (GEMV_KC / E + 63) / 64;  // 16-B vectors per lane per chunk
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  u32x4_t* xs = reinterpret_cast<u32x4_t*>(smem);  // [BT][kcv] 16-B vectors
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int K = a.K, N = a.N, B = a.B;
  const T* x = reinterpret_cast<const T*>(a.x);
  const T* W = reinterpret_cast<const T*>(a.W);
  const int row0 = (blockIdx.x * 4 + wave) * R;

  float acc[R][BT];
#pragma unroll
  for (int r = 0; r < R; ++r)
#pragma unroll
    for (int b = 0; b < BT; ++b) acc[r][b] = 0.f;

  for (int k0 = 0; k0 < K; k0 += GEMV_KC) {
    const int kl = min(GEMV_KC, K - k0);
    const int kcv = kl / E;  // vectors in this chunk
    if (k0 > 0) __syncthreads();
    // ---- stage activations (optionally LayerNorm'ed) into LDS ----
    if (a.ln_g != nullptr) {
      const T* g = reinterpret_cast<const T*>(a.ln_g);
      const T* be = reinterpret_cast<const T*>(a.ln_b);
      for (int b = wave; b < BT; b += 4) {
        float v[MAXV][E];
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < MAXV; ++i) {
          const int vi = lane + i * 64;
          if (vi < kcv && b < B) {
            unpack16<T>(*reinterpret_cast<const u32x4_t*>(x + (long long)b * a.ldx + vi * E), v[i]);
#pragma unroll
            for (int e = 0; e < E; ++e) s += v[i][e];
          }
        }
        const float mean = wave_sum(s) / (float)K;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < MAXV; ++i) {
          const int vi = lane + i * 64;
          if (vi < kcv && b < B) {
#pragma unroll
            for (int e = 0; e < E; ++e) { const float c = v[i][e] - mean; q += c * c; }
          }
        }
        const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)K + 1e-5f);
#pragma unroll
        for (int i = 0; i < MAXV; ++i) {
          const int vi = lane + i * 64;
          if (vi < kcv) {
            float o[E];
            if (b < B) {
              float gg[E], bb[E];
              unpack16<T>(*reinterpret_cast<const u32x4_t*>(g + vi * E), gg);
              unpack16<T>(*reinterpret_cast<const u32x4_t*>(be + vi * E), bb);
#pragma unroll
              for (int e = 0; e < E; ++e) o[e] = (v[i][e] - mean) * rstd * gg[e] + bb[e];
            } else {
#pragma unroll
              for (int e = 0; e < E; ++e) o[e] = 0.f;
            }
            xs[b * kcv + vi] = pack16<T>(o);
          }
        }
      }
    } else {
      for (int i = tid; i < BT * kcv; i += 256) {
        const int b = i / kcv, vi = i % kcv;
        u32x4_t v = u32x4_t{0u, 0u, 0u, 0u};
        if (b < B) v = *reinterpret_cast<const u32x4_t*>(x + (long long)b * a.ldx + k0 + vi * E);
        xs[i] = v;
      }
    }
    __syncthreads();

    // ---- stream the weight rows ----
    const int nvi = (kcv + 63) / 64;
    u32x4_t wcur[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int n = min(row0 + r, N - 1);
      wcur[r] = (lane < kcv) ? *reinterpret_cast<const u32x4_t*>(W + (long long)n * K + k0 + lane * E)
                             : u32x4_t{0u, 0u, 0u, 0u};
    }
    for (int i = 0; i < nvi; ++i) {
      const int vi = i * 64 + lane;
      u32x4_t wnext[R];
      const int vn = vi + 64;
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const int n = min(row0 + r, N - 1);
        wnext[r] = (i + 1 < nvi && vn < kcv) ? *reinterpret_cast<const u32x4_t*>(W + (long long)n * K + k0 + vn * E)
                                              : u32x4_t{0u, 0u, 0u, 0u};
      }
      if (vi < kcv) {
#pragma unroll
        for (int b = 0; b < BT; ++b) {
          const u32x4_t xv = xs[b * kcv + vi];
#pragma unroll
          for (int r = 0; r < R; ++r) acc[r][b] = dot16<T>(wcur[r], xv, acc[r][b]);
        }
      }
#pragma unroll
      for (int r = 0; r < R; ++r) wcur[r] = wnext[r];
    }
  }

  // ---- reduce across the wavefront, epilogue on lane b ----
#pragma unroll
  for (int r = 0; r < R; ++r)
#pragma unroll
    for (int b = 0; b < BT; ++b) acc[r][b] = wave_sum(acc[r][b]);

  const T* bias = reinterpret_cast<const T*>(a.bias);
  const T* res = reinterpret_cast<const T*>(a.res);
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int n = row0 + r;
    if (n >= N) continue;
    const float bv = bias ? (float)bias[n] : 0.f;
#pragma unroll
    for (int b = 0; b < BT; ++b) {
      if (lane == b && b < B) {
        float v = acc[r][b] + bv;
        if (a.gelu) v = gelu_exact(v);
        if (res) v += (float)res[(long long)b * a.ldres + n];
        if (a.y_f32) {
          a.y_f32[(long long)b * N + n] = v;
        } else if (a.kcache && n >= a.d_model) {
          const int seg = n / a.d_model;
          T* dst = reinterpret_cast<T*>(seg == 1 ? a.kcache : a.vcache);
          dst[(long long)b * a.cache_bstride + (long long)a.stt->pos * a.d_model + (n - seg * a.d_model)] = (T)v;
        } else {
          reinterpret_cast<T*>(a.y)[(long long)b * a.ldy + n] = (T)v;
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// single-query attention
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
template <typename T> __device__ __forceinline__ void store4(T* p, const float* o);
template <> __device__ __forceinline__ void store4<float>(float* p, const float* o) {
  *reinterpret_cast<f32x4_t*>(p) = f32x4_t{o[0], o[1], o[2], o[3]};
}
template <> __device__ __forceinline__ void store4<bf16_t>(bf16_t* p, const float* o) {
  typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4_t;
  bf16x4_t v;
  v[0] = (bf16_t)o[0]; v[1] = (bf16_t)o[1]; v[2] = (bf16_t)o[2]; v[3] = (bf16_t)o[3];
  *reinterpret_cast<bf16x4_t*>(p) = v;
}

__device__ __forceinline__ float group16_sum(float v) {
  v += __shfl_xor(v, 1, 64);
  v += __shfl_xor(v, 2, 64);
  v += __shfl_xor(v, 4, 64);
  v += __shfl_xor(v, 8, 64);
  return v;
}

// Streams keys t = t_begin + tg, +4, ... < t_end for lane group tg = lane>>4; each lane owns dims dq*4..+3.
// kbase/vbase point at row 0; rows are `stride` elements apart.  Produces the group-local (m, l, o[4]).
template <typename T, bool SAVE>
__device__ __forceinline__ void attend_range(const float* qv, const T* kbase, const T* vbase, long long stride,
                                             int t_begin, int t_end, int lane, float& m, float& l, float* o,
                                             float* score_out /* LDS, indexed by t */) {
  const int tg = lane >> 4, dq = lane & 15;
  m = -1.0e30f;
  l = 0.f;
  o[0] = o[1] = o[2] = o[3] = 0.f;
#pragma unroll 4
  for (int tt = t_begin; tt < t_end; tt += 4) {
    const int t = tt + tg;
    const bool valid = t < t_end;
    const int tc = valid ? t : t_end - 1;
    float kv[4], vv[4];
    load4<T>(kbase + (long long)tc * stride + dq * 4, kv);
    load4<T>(vbase + (long long)tc * stride + dq * 4, vv);
    float s = qv[0] * kv[0] + qv[1] * kv[1] + qv[2] * kv[2] + qv[3] * kv[3];
    s = group16_sum(s);
    if (SAVE) {
      if (valid && dq == 0) score_out[t] = s;
    }
    if (!valid) s = -1.0e30f;
    const float mn = fmaxf(m, s);
    const float al = expf(m - mn);
    const float p = valid ? expf(s - mn) : 0.f;
    m = mn;
    l = l * al + p;
#pragma unroll
    for (int i = 0; i < 4; ++i) o[i] = o[i] * al + p * vv[i];
  }
}

// merge the 4 lane groups of a wavefront: every lane ends with the combined (m, l, o)
__device__ __forceinline__ void merge_groups(float& m, float& l, float* o) {
  float M = fmaxf(m, __shfl_xor(m, 16, 64));
  M = fmaxf(M, __shfl_xor(M, 32, 64));
  const float sc = expf(m - M);
  l *= sc;
  l += __shfl_xor(l, 16, 64);
  l += __shfl_xor(l, 32, 64);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    float v = o[i] * sc;
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    o[i] = v;
  }
  m = M;
}

template <typename T>
__global__ __launch_bounds__(64) void dec_self_attn_kernel(const T* __restrict__ q, const T* __restrict__ kc,
                                                            const T* __restrict__ vc, long long cache_bstride,
                                                            T* __restrict__ out, int H, const DecState* __restrict__ stt) {
  const int h = blockIdx.x, b = blockIdx.y;
  const int lane = threadIdx.x;
  const int d = H * 64;
  const int n_keys = stt->pos + 1;
  float qv[4];
  load4<T>(q + (long long)b * d + h * 64 + (lane & 15) * 4, qv);
  float m, l, o[4];
  attend_range<T, false>(qv, kc + (long long)b * cache_bstride + h * 64, vc + (long long)b * cache_bstride + h * 64, d, 0,
                         n_keys, lane, m, l, o, nullptr);
  merge_groups(m, l, o);
  if (lane < 16) {
    const float inv = 1.0f / l;
    float r[4] = {o[0] * inv, o[1] * inv, o[2] * inv, o[3] * inv};
    store4<T>(out + (long long)b * d + h * 64 + lane * 4, r);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void dec_cross_attn_kernel(const T* __restrict__ q, const T* __restrict__ ck,
                                                              const T* __restrict__ cv, T* __restrict__ out, int H,
                                                              int Tlen, const int* __restrict__ align_slot,
                                                              float* __restrict__ align, int Ha, int P,
                                                              const DecState* __restrict__ stt) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* sc = reinterpret_cast<float*>(smem);  // [Tlen] raw scores (alignment heads only)
  __shared__ float wm[4], wl[4], wo[4][64];
  const int h = blockIdx.x, b = blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int d = H * 64;
  const int slot = align_slot ? align_slot[h] : -1;
  float qv[4];
  load4<T>(q + (long long)b * d + h * 64 + (lane & 15) * 4, qv);
  const long long base = ((long long)b * H + h) * Tlen * 64;
  int per = (Tlen + 3) / 4;
  per = (per + 3) & ~3;
  const int t0 = min(wave * per, Tlen), t1 = min(t0 + per, Tlen);
  float m, l, o[4];
  if (slot >= 0)
    attend_range<T, true>(qv, ck + base, cv + base, 64, t0, t1, lane, m, l, o, sc);
  else
    attend_range<T, false>(qv, ck + base, cv + base, 64, t0, t1, lane, m, l, o, nullptr);
  merge_groups(m, l, o);
  if (lane == 0) { wm[wave] = m; wl[wave] = l; }
  if (lane < 16) {
#pragma unroll
    for (int i = 0; i < 4; ++i) wo[wave][lane * 4 + i] = o[i];
  }
  __syncthreads();
  const float M = fmaxf(fmaxf(wm[0], wm[1]), fmaxf(wm[2], wm[3]));
  const float s0 = expf(wm[0] - M), s1 = expf(wm[1] - M), s2 = expf(wm[2] - M), s3 = expf(wm[3] - M);
  const float L = wl[0] * s0 + wl[1] * s1 + wl[2] * s2 + wl[3] * s3;
  const float inv = 1.0f / L;
  if (tid < 64) {
    const float v = (wo[0][tid] * s0 + wo[1][tid] * s1 + wo[2][tid] * s2 + wo[3][tid] * s3) * inv;
    out[(long long)b * d + h * 64 + tid] = (T)v;
  }
  if (slot >= 0) {
    float* row = align + (((long long)b * Ha + slot) * P + stt->pos) * Tlen;
    for (int t = tid; t < Tlen; t += 256) row[t] = expf(sc[t] - M) * inv;
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
  // ---- pass 1: best text token, best timestamp token ----
  MaxIdx bt{-INFINITY, 0x7fffffff}, bs{-INFINITY, 0x7fffffff};
  for (int v = tid; v < V; v += 1024) {
    const float s = score(v);
    MaxIdx c{s, v};
    if (v < ts_begin) bt = better(bt, c); else bs = better(bs, c);
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
    for (int v = ts_begin + tid; v < V; v += 1024) {
      const float s = score(v);
      sum += expf(s - bs.v);
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
  if (a.K % E != 0 || (a.ln_g && a.K > GEMV_KC) || a.B > BT) return hipErrorInvalidValue;
  const int kc = a.K < GEMV_KC ? a.K : GEMV_KC;
  const size_t lds = (size_t)BT * kc * sizeof(T);
  // rows per wavefront: keep >= ~2000 wavefronts in flight when N allows it
  int R = 1;
  if (a.N >= 16384) R = 4;
  else if (a.N >= 3072) R = 2;
  const int rows_per_block = 4 * R;
  dim3 grid((a.N + rows_per_block - 1) / rows_per_block);
  if (R == 4) hipLaunchKernelGGL((gemv_kernel<T, BT, 4>), grid, dim3(256), lds, st, a);
  else if (R == 2) hipLaunchKernelGGL((gemv_kernel<T, BT, 2>), grid, dim3(256), lds, st, a);
  else hipLaunchKernelGGL((gemv_kernel<T, BT, 1>), grid, dim3(256), lds, st, a);
  return hipGetLastError();
}

template <typename T, int BT>
static hipError_t gemv_attr() {
  // the f32 / 16-stream variants stage up to 80 KiB of activations: raise the dynamic-LDS cap once
  const int cap = 96 * 1024;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemv_kernel<T, BT, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, cap);
  if (e != hipSuccess) return e;
  e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemv_kernel<T, BT, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, cap);
  if (e != hipSuccess) return e;
  return hipFuncSetAttribute(reinterpret_cast<const void*>(&gemv_kernel<T, BT, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, cap);
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

template <typename T>
static hipError_t gemv_b(const GemvArgs& a, hipStream_t st) {
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
                                void* out, int B, int H, const DecState* stt, hipStream_t st) {
  if (dtype == 1)
    hipLaunchKernelGGL(dec_self_attn_kernel<bf16_t>, dim3(H, B), dim3(64), 0, st, (const bf16_t*)q, (const bf16_t*)kc,
                       (const bf16_t*)vc, cache_bstride, (bf16_t*)out, H, stt);
  else
    hipLaunchKernelGGL(dec_self_attn_kernel<float>, dim3(H, B), dim3(64), 0, st, (const float*)q, (const float*)kc,
                       (const float*)vc, cache_bstride, (float*)out, H, stt);
  return hipGetLastError();
}

hipError_t launch_dec_cross_attn(int dtype, const void* q, const void* ck, const void* cv, void* out, int B, int H,
                                 int T, const int* align_slot_for_head, float* align, int Ha, int P,
                                 const DecState* stt, hipStream_t st) {
  const size_t lds = (size_t)T * sizeof(float);
  if (dtype == 1)
    hipLaunchKernelGGL(dec_cross_attn_kernel<bf16_t>, dim3(H, B), dim3(256), lds, st, (const bf16_t*)q,
                       (const bf16_t*)ck, (const bf16_t*)cv, (bf16_t*)out, H, T, align_slot_for_head, align, Ha, P, stt);
  else
    hipLaunchKernelGGL(dec_cross_attn_kernel<float>, dim3(H, B), dim3(256), lds, st, (const float*)q, (const float*)ck,
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
