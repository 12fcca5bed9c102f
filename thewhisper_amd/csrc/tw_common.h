// Internal declarations shared by the gfx950 kernel translation units and the C-ABI layer.
// MI355X (gfx950, CDNA4) only: wave = 64 lanes, MFMA 16x16x32 bf16 / 16x16x4 f32.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __bf16 bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
// float16 contexts (TW_F16, round 4): the reference's streaming default dtype (R:thestage_speechkit/streaming/streaming_pipeline.py:369-370).
// Same 16-bit containers, same MFMA rate (v_mfma_f32_16x16x32_f16), 10 mantissa bits instead of 7.
typedef _Float16 f16_t;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8_t;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;  // a 16-B register quad (native vector: stays in VGPRs)
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2_t;

// ---- cross-lane reductions without LDS traffic (gfx950) ----------------------------------------------------------
// Inside a 16-lane DPP row: v_add/v_max_f32_dpp with quad_perm xor-1, xor-2, row_ror:4, row_ror:8.  Across rows:
// v_permlane16_swap / v_permlane32_swap with both operands = v leave {v.rowA, v.rowA..} / {v.rowB, v.rowB..} in the two
// registers, so their sum (max) is the xor-16 / xor-32 butterfly step.  (A __shfl_xor butterfly is six dependent
// ds_bpermute round trips through the LDS crossbar.)  The swaps are issued through inline asm: with hipcc 7.2 the
// __builtin_amdgcn_permlane{16,32}_swap builtins fed with one value combine the two results as r[0]+r[0].
template <int CTRL> __device__ __forceinline__ float tw_dpp(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ void tw_swap16(float& a, float& b) { asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b)); }
__device__ __forceinline__ void tw_swap32(float& a, float& b) { asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b)); }
__device__ __forceinline__ float tw_row16_sum(float v) {
  v += tw_dpp<0xB1>(v);   // quad_perm [1,0,3,2]
  v += tw_dpp<0x4E>(v);   // quad_perm [2,3,0,1]
  v += tw_dpp<0x124>(v);  // row_ror:4
  v += tw_dpp<0x128>(v);  // row_ror:8
  return v;
}
__device__ __forceinline__ float tw_row16_max(float v) {
  v = fmaxf(v, tw_dpp<0xB1>(v));
  v = fmaxf(v, tw_dpp<0x4E>(v));
  v = fmaxf(v, tw_dpp<0x124>(v));
  v = fmaxf(v, tw_dpp<0x128>(v));
  return v;
}
__device__ __forceinline__ float tw_xor16_sum(float v) { float a = v, b = v; tw_swap16(a, b); return a + b; }
__device__ __forceinline__ float tw_xor32_sum(float v) { float a = v, b = v; tw_swap32(a, b); return a + b; }
__device__ __forceinline__ float tw_xor16_max(float v) { float a = v, b = v; tw_swap16(a, b); return fmaxf(a, b); }
__device__ __forceinline__ float tw_xor32_max(float v) { float a = v, b = v; tw_swap32(a, b); return fmaxf(a, b); }
__device__ __forceinline__ float tw_wave_sum(float v) { return tw_xor32_sum(tw_xor16_sum(tw_row16_sum(v))); }

// Folded-LayerNorm scalars and their application, with the fused operations spelled out.  Left to the compiler, `ss * inv_k - mean * mean`
// becomes fma(ss, inv_k, -(mean * mean)) in one template instantiation and fma(-mean, mean, ss * inv_k) in another (seen in the ISA of the
// one-tile and the several-tiles MXFP8 projection kernels, round 6): results a few float32 ulps apart, i.e. a stream's bits would depend
// on which kernel flavour the number of rows in its launch selects.
__device__ __forceinline__ void tw_ln_scalars(float s, float ss, float inv_k, float& mean, float& rstd) {
  mean = s * inv_k;
  const float m2 = mean * mean;
  rstd = __frsqrt_rn(fmaxf(fmaf(ss, inv_k, -m2), 0.f) + 1e-5f);
}
// (sum, sum of squares) of four consecutive columns in ONE order.  The per-(row, 32-column block) statistics the encoder GEMMs leave for
// the folded LayerNorm of their consumer are built from these by a fixed tree - ((P0 + P1) + (P2 + P3)) + ((P4 + P5) + (P6 + P7)) over the
// block's eight groups of four columns - by BOTH epilogues of k_gemm.hip (a lane of the plain epilogue holds groups q and 4 + q, a lane
// of the LDS-staged one groups 2j and 2j + 1): until round 6 the two summed in different orders, i.e. a clip's encoder states depended
// on whether its pass was large enough for the large-M kernel (tools/dbg/batch_invariance.py).
__device__ __forceinline__ void tw_stat4(float x0, float x1, float x2, float x3, float& s, float& ss) {
  s = ((x0 + x1) + x2) + x3;
  ss = fmaf(x3, x3, fmaf(x2, x2, fmaf(x1, x1, x0 * x0)));
}
__device__ __forceinline__ float tw_ln_apply(float v, float mean, float rstd, float gw, float cb) { return fmaf(rstd, fmaf(-mean, gw, v), cb); }
__device__ __forceinline__ float tw_wave_max(float v) { return tw_xor32_max(tw_xor16_max(tw_row16_max(v))); }

template <typename T> struct ElemTraits;
template <> struct ElemTraits<float> { static constexpr int kPer16B = 4; static constexpr int kCode = 0; };
template <> struct ElemTraits<bf16_t> { static constexpr int kPer16B = 8; static constexpr int kCode = 1; };
template <> struct ElemTraits<f16_t> { static constexpr int kPer16B = 8; static constexpr int kCode = 2; };

// ---- the 16-bit element types share everything but three instructions: the MFMA, the float -> T pair conversion, the 2-way dot ----
template <typename T> struct Pair16;
template <> struct Pair16<bf16_t> { typedef bf16x2_t x2; typedef bf16x8_t x8; };
template <> struct Pair16<f16_t> { typedef f16x2_t x2; typedef f16x8_t x8; };
// acc += A(16 x 32) . B(32 x 16) on 16-B operand fragments of T
template <typename T> __device__ __forceinline__ f32x4_t tw_mfma32(const u32x4_t& a, const u32x4_t& b, f32x4_t acc);
template <> __device__ __forceinline__ f32x4_t tw_mfma32<bf16_t>(const u32x4_t& a, const u32x4_t& b, f32x4_t acc) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), acc, 0, 0, 0);
}
template <> __device__ __forceinline__ f32x4_t tw_mfma32<f16_t>(const u32x4_t& a, const u32x4_t& b, f32x4_t acc) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), acc, 0, 0, 0);
}
// two floats -> one dword holding (T)lo | (T)hi << 16 (round to nearest even; float16: saturating at the largest finite value,
// as HF clamps its float16 hidden states, HF:models/whisper/modeling_whisper.py:409-411)
template <typename T> __device__ __forceinline__ unsigned tw_pack2(float lo, float hi) {
  typename Pair16<T>::x2 v;
  if constexpr (sizeof(T) == 2 && ElemTraits<T>::kCode == 2) {
    lo = fminf(fmaxf(lo, -65504.f), 65504.f);
    hi = fminf(fmaxf(hi, -65504.f), 65504.f);
  }
  v[0] = (T)lo;
  v[1] = (T)hi;
  return __builtin_bit_cast(unsigned, v);
}
// float -> T for activations (float16: saturating, see tw_pack2)
template <typename T> __device__ __forceinline__ T tw_cast(float v) {
  if constexpr (ElemTraits<T>::kCode == 2) v = fminf(fmaxf(v, -65504.f), 65504.f);
  return (T)v;
}
// the same without the saturation, for values known to be in range (probabilities, convex combinations of stored values)
template <typename T> __device__ __forceinline__ unsigned tw_pack2_inrange(float lo, float hi) {
  typename Pair16<T>::x2 v;
  v[0] = (T)lo;
  v[1] = (T)hi;
  return __builtin_bit_cast(unsigned, v);
}
// s += a0 * b0 + a1 * b1 on pairs of T, float accumulate
template <typename T> __device__ __forceinline__ float tw_dot2(typename Pair16<T>::x2 a, typename Pair16<T>::x2 b, float s);
template <> __device__ __forceinline__ float tw_dot2<bf16_t>(bf16x2_t a, bf16x2_t b, float s) { return __builtin_amdgcn_fdot2_f32_bf16(a, b, s, false); }
template <> __device__ __forceinline__ float tw_dot2<f16_t>(f16x2_t a, f16x2_t b, float s) { return __builtin_amdgcn_fdot2(a, b, s, false); }

// launcher-side dispatch on a context element type code (0 f32, 1 bf16, 2 f16)
#define TW_DISPATCH3(code, T, ...)                       \
  do {                                                   \
    if ((code) == 1) { typedef bf16_t T; __VA_ARGS__; }  \
    else if ((code) == 2) { typedef f16_t T; __VA_ARGS__; } \
    else { typedef float T; __VA_ARGS__; }               \
  } while (0)

// Fragment-major ("xt") layout of the decoder's per-token activations, a [16 streams][K] block stored as the MFMA B
// operand of the decode projections reads it: 64-B step s = k / (4E), then lane = ((k / E) & 3) * 16 + stream, then the
// E elements of that lane's 16-B vector.  One wavefront request for step s is 1 KiB contiguous.  Buffers hold whole groups
// of 16 streams (group g of a [.., K] buffer starts at element g*16*K; streams >= B are never written nor read back).
template <typename T>
__device__ __forceinline__ long long tw_xt_index(int stream, int k) {
  constexpr int E = ElemTraits<T>::kPer16B;
  return ((long long)(k / (4 * E)) * 64 + ((k / E) & 3) * 16 + stream) * E + (k % E);
}

// Decoder K / V caches, per (stream, head), fragment-major for the single-query MFMA attention (k_decode.hip):
//  K  : key tile t/16, dim step c/(4E): lane = ((c/E)&3)*16 + t%16 holds dims of one 16-B vector      (A operand of Q.K^T)
//  V^T: key step t/(4E), dim tile c/16: lane = ((t/E)&3)*16 + c%16 holds E consecutive KEYS of dim c  (A operand of P.V)
// 64 keys of either operand are one contiguous run of 64*64 elements.
template <typename T>
__device__ __forceinline__ long long tw_kf_index(int t, int c) {
  constexpr int E = ElemTraits<T>::kPer16B;
  return ((long long)((t >> 4) * (64 / (4 * E)) + c / (4 * E)) * 64 + ((c / E) & 3) * 16 + (t & 15)) * E + (c % E);
}
template <typename T>
__device__ __forceinline__ long long tw_vtf_index(int t, int c) {
  constexpr int E = ElemTraits<T>::kPer16B;
  return ((long long)((t / (4 * E)) * 4 + (c >> 4)) * 64 + ((t / E) & 3) * 16 + (c & 15)) * E + (t % E);
}

// fp8 (e4m3) variants of the cross-attention caches (TW_BF16_MXFP8 contexts), byte offsets inside one (stream, head) slab of
// Tp*64 bytes; every key carries one power-of-two scale byte per head (value = fp8 * 2^(sb - 127), the rule of sk_quant_mx8):
//  K  : 16-key tile t/16 = 1 KiB, lane = (c/16)*16 + t%16 holds dims 16*(c/16) .. +15 of key t          (A operand of Q.K^T)
//  V^T: 64-key group, dim tile c/16 = 1 KiB, lane = ((t%64)/16)*16 + c%16 holds 16 consecutive keys of dim c   (A of P.V)
__device__ __forceinline__ long long tw_kf8_index(int t, int c) {
  return ((long long)(t >> 4) * 64 + (c >> 4) * 16 + (t & 15)) * 16 + (c & 15);
}
__device__ __forceinline__ long long tw_vtf8_index(int t, int c) {
  return (((long long)(t >> 6) * 4 + (c >> 4)) * 64 + ((t & 63) >> 4) * 16 + (c & 15)) * 16 + (t & 15);
}

// Affine map from a logical GEMM row m to an element offset:  (m / rpb) * bstride + (m % rpb) * rstride.
// Lets one GEMM read convolution windows as overlapping rows of a padded token-major buffer and
// write into padded / per-batch layouts without im2col copies.
struct RowMap {
  int rpb;            // rows per batch block (>= M for a plain matrix)
  long long bstride;  // elements between batch blocks
  long long rstride;  // elements between consecutive rows
};
static inline RowMap plain_rows(long long ld) { return RowMap{1 << 30, 0, ld}; }

// GEMM epilogue descriptor: v = acc + bias[n]; GELU?; + res[rmap(m % res_mod) + n]; store by mode.
enum { EPI_ROWMAJOR = 0, EPI_HEADSPLIT = 1, EPI_QKV_ENC = 2, EPI_KV_CROSS = 3, EPI_KV_CROSS8 = 4 };
struct GemmEpilogue {
  const void* bias;   // [N] in T, may be null
  int gelu;           // exact erf GELU
  const void* res;    // residual, element type T, may be null
  RowMap res_map;     // row map of the residual
  int res_mod;        // residual row = m % res_mod (positional table broadcast); 0 = m
  int mode;           // EPI_*
  // EPI_ROWMAJOR: out + cmap(m) + n
  RowMap c_map;
  // head-split modes: rows m = b*T + t.  out index ((b*H + h)*T + t)*64 + dd
  int T;              // tokens per batch entry
  int Tp;             // padded T of transposed V
  int H;
  void* out;          // primary output (q / k of cross)
  void* out2;         // k (QKV_ENC) or v (KV_CROSS)
  void* out3;         // v transposed (QKV_ENC); KV_CROSS8: per-key scale bytes of K [B][H][Tp]
  void* out4;         // KV_CROSS8: per-key scale bytes of V^T [B][H][Tp]
  // Folded pre-LayerNorm of the ENCODER (round 4; the decoder's projections have had it since round 1, k_decode.hip).
  // Producer side (a GEMM that writes the residual stream, EPI_ROWMAJOR): stats_out[m][n / 32] = (sum, sum of squares) of the 32
  // stored values of row m in that column block - the LayerNorm statistics of the row in N / 32 partials, fixed order, no atomics.
  // Consumer side (the QKV / fc1 GEMM, whose weight carries the LayerNorm gain, tw_finalize_weights): the A operand is the RAW
  // residual stream, y = rstd (acc - mean gw[n]) + cb[n] with (mean, rstd) of row m from stats_in[m][0 .. parts); `bias` is then null
  // (cb contains it).  No normalised copy of the stream is written or read and two launches per layer disappear.
  float* stats_out;
  int staged;   // set by launch_gemm: the large-M kernel stages row-contiguous 16-bit outputs through LDS (gemm_epilogue_staged)
  const float* stats_in; int stats_in_parts;
  const float* ln_gw; const float* ln_cb;
};

// ---- launchers (one per kernel family); all enqueue on `st` and return hipGetLastError() ----
// C[M,N] = epi(A[M,K] * W[N,K]^T).  T = float or bf16.  K % 64 == 0 (bf16) / % 32 == 0 (f32), N % 64 == 0.
hipError_t launch_gemm(int dtype, const void* A, RowMap amap, const void* W, int M, int N, int K,
                       const GemmEpilogue& ep, hipStream_t st);

hipError_t launch_layernorm(int dtype, const void* x, const void* g, const void* b, void* y, int rows, int d,
                            hipStream_t st);
// mel [B, n_mels, F] (src dtype) -> melT [B, F+2, C] (ctx dtype), zero pad rows 0 and F+1 and channels >= n_mels
hipError_t launch_mel_transpose(int dst_dtype, int src_dtype, const void* mel, void* melT, int B, int n_mels, int F,
                                int C, hipStream_t st);
hipError_t launch_convert(int dst_dtype, int src_dtype, const void* src, void* dst, long long n, float scale,
                          hipStream_t st);
hipError_t launch_fill_zero(void* dst, long long bytes, hipStream_t st);
// conv weight [co][ci][3] (src dtype) -> [co][3][C] (dst dtype), zero for ci >= n_ci
hipError_t launch_conv_weight_reorder(int dst_dtype, int src_dtype, const void* src, void* dst, int co, int ci, int C,
                                      hipStream_t st);
// positional table [n_old, d] f32-> [n_new, d] ctx dtype by linear interpolation (align_corners=False)
hipError_t launch_interp_positions(int dst_dtype, int src_dtype, const void* src, void* dst, int n_old, int n_new,
                                   int d, hipStream_t st);

// log-mel (A1)
struct LogmelTables {
  const double* twiddle;  // cos(2*pi*j/400), j in [0,400)
  const double* window;   // periodic hann(400)
  const float* bank;      // [201][n_mels] f32 (HF casts the f64 bank to f32)
  const int* lo;          // per mel: first non-zero bin
  const int* hi;          // per mel: one past last non-zero bin
};
hipError_t launch_logmel(const float* pcm, long long pcm_stride, const int* n_valid_dev, int B, int n_samples,
                         int n_mels, const LogmelTables& tb, float* logspec_ws, unsigned* max_ws, void* out,
                         int out_dtype, hipStream_t st);

// encoder self-attention (non-causal, head_dim 64).  q,k: [B,H,T,64]; vt: [B,H,64,Tp]; out: [B*T, H*64]
hipError_t launch_enc_attention(int dtype, const void* q, const void* k, const void* vt, void* out, int B, int H,
                                int T, int Tp, hipStream_t st);

// ---- decoder (single new token per stream) ----
struct DecState {   // lives in device memory so that a captured graph is position independent
  int pos;          // position of the token being consumed this step
  int n_prompt;     // begin_index
  int step_limit;   // last position that may be produced (max_len - 1)
  int unfinished;   // count of unfinished streams (written by the sampler)
};
// "Rows mode" of the decoder kernels (batched prefill of FORCED tokens, api.hip: prefill_core): the B rows of a launch are not B
// streams at one position but rs streams x B / rs consecutive positions - row r is stream r % rs at position pos + r / rs (stream
// fastest: a group of 16 rows is one position of 16 streams).  Row-wise work (projections, LayerNorm statistics, the fused cross
// query) does not care; what does is everything that touches a stream's caches or its position: the embedding's positional row,
// the K / V^T cache scatter of the QKV projection, the self-attention's key range (causal inside the launch: keys [0, p]), the
// cross-attention's arena and its alignment row.  rs = 0: the ordinary step (row = stream, one position).
__device__ __forceinline__ void tw_row_of(int r, int rs, int pos, int& stream, int& p) {
  if (rs > 0) {
    stream = r % rs;
    p = pos + r / rs;
  } else {
    stream = r;
    p = pos;
  }
}
// x[b] = tok_emb[ids[b]] + pos_emb[pos]   (rows mode: pos_emb[pos + b / rows_streams])
hipError_t launch_embed(int dtype, const int* ids, const DecState* stt, const void* tok, const void* pos, void* x,
                        int B, int d, int rows_streams, hipStream_t st);
// y[b, n] = epi( LN?(x[b,:]) . W[n,:] + bias[n] )  for b < B <= 16.
struct GemvArgs {
  const void* x; int ldx;        // [16, K] input (T), fragment-major (tw_xt_index); ldx unused
  // folded pre-LayerNorm (see k_decode.hip): W already carries the gain, ln_gw[n] = sum_k g[k] W[n,k],
  // ln_cb[n] = sum_k beta[k] W[n,k] + bias[n]  (float32, null = no LayerNorm); requires K <= 1280
  const float* ln_gw; const float* ln_cb;
  const void* W; const void* bias;     // [N, K] in the fragment-major layout of launch_tile_weights, [N]
  const unsigned char* wscale;         // non-null: W is MXFP8 (launch_quant_mx8), these are its block scales; bf16 activations
  int a16;                             // with wscale: 1 = W8A16 (weights widened to bf16 in registers, activations not quantised)
  int N, K, B;
  int gelu;
  const void* res; int ldres;    // residual [16, N] (fragment-major) added after bias/act; output then fragment-major too
  void* y; int ldy;              // output (T): fragment-major [16, N] with res or gelu, else row-major [B, ldy]; or
  float* y_f32;                  // float32 output [B, N] (logits) when non-null
  // optional KV-cache scatter for the fused self-attention QKV projection: rows [d,2d) -> kcache, [2d,3d) -> vcache
  // (fragment-major per (stream, head): tw_kf_index / tw_vtf_index; cache_bstride = elements per stream = H * rows * 64,
  //  cache_hstride = elements per (stream, head) = rows * 64 - handed over so that the epilogue has no 64-bit division to do)
  void* kcache; void* vcache; long long cache_bstride; long long cache_hstride; int d_model; const DecState* stt;
  int rg;  // skinny path: 16-row tiles walked per workgroup (set by the launcher)
  int tr;  // weight rows per workgroup tile the weights were laid out for by launch_tile_weights (0 / 16, 8 or 4)
  // "cross query ahead" (api.hip: decode_core): float32 [B][d_model] pre-activation of the NEXT LayerNorm'd projection.
  //  * QKV launch (kcache set): rows [3d,4d) of W are that projection's folded weight; x . W^T + ln_cb is stored here;
  //  * residual launch (res set, nsplit = d_model < N): rows [nsplit,N) of W are the composed matrix (projection . this one);
  //    they add their product to u in place, rows [0,nsplit) produce the residual stream as usual.
  //    They also leave (sum, sum of squares) of the residual rows they produced in stats[stream][tile][2] (tile = n / tr):
  //    the consumer of u applies the folded LayerNorm with them.
  float* u; int nsplit; float* stats;
  int rows_streams;  // rows mode (tw_row_of): > 0 = number of streams the B rows cycle through; only the K / V^T scatter uses it
};
hipError_t launch_gemv(int dtype, const GemvArgs& a, hipStream_t st);
hipError_t init_decode_kernels();
// row-major [N][K] -> the fragment-major layout launch_gemv reads (k_decode.hip); dst holds ceil(N/16)*16 rows
hipError_t launch_tile_weights(int dtype, const void* src, void* dst, int N, int K, int tr, hipStream_t st);
// row-major bf16 [N][K] -> MXFP8 fragments (ceil(N/16)*16*K bytes) + block scales (ceil(N/16)*16*K/32 bytes); K % 128 == 0
hipError_t launch_quant_mx8(const void* src_bf16, void* dst_fp8, void* dst_scales, int N, int K, int tr, hipStream_t st);   // tr: 16 or 8 rows per tile
// weight preparation for the folded pre-LayerNorm: W <- Wsrc * g (may alias), gw, cb as above
hipError_t launch_fold_ln(int dtype, void* W, const void* Wsrc, const void* g, const void* beta, const void* bias, float* gw,
                          float* cb, int N, int K, hipStream_t st);  // once per process: dynamic-LDS caps of the gemv instantiations
// self attention over the growing cache: q [B,d] row-major; kc/vc per (stream, head) `rows` (multiple of 64) keys, fragment-major
// (tw_kf_index / tw_vtf_index); out [16,d] fragment-major (tw_xt_index).
// key_bound: host-known upper bound of pos+1 for this call; <= 256 selects the single-round-trip kernel
hipError_t launch_dec_self_attn(int dtype, const void* q, const void* kc, const void* vc, int rows, void* out, int B, int H,
                                int key_bound, const DecState* stt, int rows_streams, hipStream_t st);
// cross attention over cached encoder K/V: ck/cv per (stream, head) Tp keys fragment-major; out fragment-major; align rows:
// for head h with align_slot[h] >= 0 write the softmax row to align[((b*Ha + slot)*P + pos)*T + t]
// ksc / vsc non-null: ck / cv are the fp8 caches above (bf16 contexts only), these their per-key scale bytes [B][H][Tp]
// Query operand: either q [B,d] row-major (T), or - fq.u non-null - the un-normalised pre-activation u of the folded
// LayerNorm'd query projection: q = rstd (u - mean gw) + cb with the statistics of x's row taken inside the launch.
struct FusedQ {
  const float* u;    // [B][d] float32
  const float* stats; int n_part;   // [B][n_part][2]: partial (sum, sum of squares) of the residual row the LayerNorm normalises
  const float* gw; const float* cb;   // folded-LayerNorm companions of the query projection (float32 [d])
  int d;
};
hipError_t launch_dec_cross_attn(int dtype, const void* q, const FusedQ& fq, const void* ck, const void* cv, void* out, int B, int H,
                                 int T, int Tp, const int* align_slot_for_head, float* align, int Ha, int P,
                                 const DecState* stt, const unsigned char* ksc, const unsigned char* vsc, int rows_streams,
                                 hipStream_t st);
// load-time composition for "cross query ahead": out[n][k] = sum_j A[n][j] Bm[j][k] (row-major, T), c0[n] = sum_j A[n][j] bvec[j]
hipError_t launch_compose(int dtype, const void* A, const void* Bm, const void* bvec, void* out, float* c0, int N, int J, int K,
                          hipStream_t st);

struct SamplerPartial { float bt_v; int bt_i; float bs_v; int bs_i; float sum; int pad_[3]; };  // per vocabulary slice
struct SamplerArgs {   // A10 + argmax + bookkeeping
  const float* logits; int V; int B;
  int* seq; int seq_ld;          // [B, seq_ld] token history (prompt included)
  int* cur_ids;                  // [B] token to feed next step
  int* finished;                 // [B]
  int* last_ts;                  // [B] last timestamp token sampled (-1 none)
  DecState* stt;
  int eos, pad, min_new, timestamps, no_ts_id, max_initial_ts;  // max_initial_ts < 0: unset
  const int* begin_suppress; int n_begin_suppress;
  const unsigned* suppress_bits;  // static suppress list as a V-bit map (launch_suppress_bitmap)
  SamplerPartial* partials;       // [B][32] workspace between the two sampler launches
  int n_slices;                   // vocabulary slices per stream (set by launch_sampler)
  // A6 fused into the sampler's last launch (round 5): x_next non-null = the wavefront that has just chosen stream b's token also
  // writes the NEXT step's input row, tok_emb[token] + pos_emb[pos + 1], in the fragment-major activation layout (what launch_embed
  // would write at the start of that step: one launch per step less).  dtype = the context's element type code (0 f32, 1 bf16, 2 f16).
  const void* tok_emb; const void* pos_emb; void* x_next; int d; int dtype;
  // Rows mode (draft-and-verify, api.hip: verify_round): the B rows of `logits` are rows_streams streams x B / rows_streams
  // consecutive positions starting at row_pos0 (host-known), row r = stream r % rows_streams at position row_pos0 + r / rows_streams, and
  // every row is sampled with the state the ordinary loop would have had at that position IF the given tokens (seq) are what it had
  // produced so far: history = seq[stream][0 .. p], last timestamp token = row_lastts[r] (host-computed), nothing finished.
  int rows_streams; int row_pos0;
  const int* row_lastts;     // [B] per row: last timestamp token among the generated tokens at positions <= p (-1 none)
  int* row_choice;           // [B] out: the token the loop would have produced at position p + 1
  int row_last_pos;          // last position of the ROUND (a round may take several launches): rows at p < row_last_pos are compared with seq[p + 1]
  int row_is_last;           // this launch ends the round: without a mismatch the round's result is the token after row_last_pos
  int* verify_state;         // [2 + 64]: [0] = first position whose choice differs from the given next token (INT_MAX none; atomicMin),
                             //           [1] = p_acc + 1 once the round is decided (else 0): the device position, [2 + b] = token at p_acc + 1
};
hipError_t launch_sampler(const SamplerArgs& a, hipStream_t st);   // sampler + pos advance
// rows mode: sampler over every row + the round's decision (see SamplerArgs); the position is NOT advanced unless the round is decided
hipError_t launch_sampler_rows(const SamplerArgs& a, hipStream_t st);
hipError_t launch_suppress_bitmap(const int* list, int n, unsigned* bits, int V, hipStream_t st);  // zero + set bits
hipError_t launch_advance(DecState* stt, int n, hipStream_t st);     // pos += n only (teacher-forced stepping, prefill)

// A11: alignment rows -> token timestamps
struct DtwArgs {
  const float* align; int Ha; int P; int T;   // [B][Ha][P][T]
  int B; int n_prompt; int n_rows;            // rows used: [n_prompt, n_rows)
  const int* n_cols;                          // [B] columns kept (num_frames // 2), device
  int median_width;
  float* zbuf;     // [B][Ha][N][T] workspace
  float* mat;      // [B][N][T] workspace
  signed char* trace;  // [B][(N+1)*(T+1)]
  float* out_ts;   // [B][n_rows + 1]
  double time_precision;
};
hipError_t launch_token_timestamps(const DtwArgs& a, hipStream_t st);

// 8f-2: energy voice-activity gate, one wavefront per stream, `n_frames` consecutive 512-sample frames each (k_vad.hip)
hipError_t launch_vad_energy(const float* pcm, long long stream_stride, int B, int n_frames, float* state, float* prob,
                             hipStream_t st);
