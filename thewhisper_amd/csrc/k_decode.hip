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
// (stamp 0 is taken before the launch's position is known: it is kept in a register and stored with stamp 1)
#define TW_TS(k) do { if ((k) == 0) probe_t0 = wall_clock64(); else if (threadIdx.x == 0) { unsigned long long* q_ = g_probe_ts + ((size_t)cur_pos * 1024 + blockIdx.x) * 8; \
                                                                                         if ((k) == 1) q_[0] = probe_t0; q_[k] = wall_clock64(); } } while (0)
#else
#define TW_TS(k) do { } while (0)
#endif

// Compile-time switches of the projection kernel (defaults = what ships; tools/dbg/probe_gemv.hip builds the other settings for A/B runs):
//  TW_RED_STRIDE   row stride (floats) of the partial tiles in LDS.  A partial tile is written [weight row][stream] by the MFMA lanes and
//                  read [stream][weight row] by the epilogue threads (16 consecutive rows of one stream per 16 threads, for the stores):
//                  with 16 floats per row a 32-lane group of a ds_read_b32 hit 4 banks, 8 addresses each (8-way conflict on every read
//                  of the reduction); 17 spreads it over all 32 banks, the writers stay at <= 2 addresses per bank (free for stores).
//  TW_CG_ORDER     several groups of 16 streams per launch: the HBM weight requests leave FIRST and the (L2) activation requests follow in
//                  consumption order (step-major); 0 = the order of the one-group kernel (activations first, group-major), which puts
//                  20 KiB of L2 traffic per wavefront in front of the first HBM request.
//  TW_CG_EPI_ALL   ... the epilogue of the groups is spread over all 512 threads (two halves of the workgroup take alternate groups).
//  TW_CG_RING     several groups, 16-bit or f32 weights, one tile per workgroup: the template parameter SK_MAXS is then the wavefront's
//                  WHOLE step count and the operands go through two register RINGS (weights: 5 steps deep, activations: 3 steps x
//                  CG groups) refilled as soon as a step has been contracted - straight-line code, the compiler's vmcnt counts are
//                  exact.  The point is registers: the one-round scheme holds CG x 5 activation fragments at once (80 registers at 64
//                  streams, 134-158 in total: ONE 512-thread workgroup per CU, so the 320-workgroup launches ran as two rounds on
//                  the decode loop's 160 CUs); the rings need 48 and two workgroups fit.
#ifndef TW_RED_STRIDE
#define TW_RED_STRIDE 17
#endif
#ifndef TW_CG_RING
#define TW_CG_RING 1
#endif
#ifndef TW_CG_ORDER
#define TW_CG_ORDER 1
#endif
#ifndef TW_CG_EPI_ALL
#define TW_CG_EPI_ALL 1
#endif
constexpr int kRedTile = 16 * TW_RED_STRIDE;   // floats per partial tile in LDS

__device__ __forceinline__ float wave_sum(float v) { return tw_wave_sum(v); }
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
  const float e = fmaf(-(p * t), __expf(-az * az), 1.0f);   // (spelled out: one rounding sequence in every instantiation)
  return 0.5f * x * (1.0f + copysignf(e, z));
}

// NB for every helper below that takes bf16 pairs out of a 16-byte vector: hipcc (ROCm 7.2) folds `bit_cast<bf16x2>(v[i])`
// inside an unrolled loop to element 0 for every i (seen in the ISA: identical instructions), so pairs are always taken
// with named __builtin_shufflevector selections of the whole bf16x8 vector.

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
template <> __device__ __forceinline__ u32x4_t pack16<f16_t>(const float* in) {   // probabilities: in range, no saturation needed
  f16x8_t f;
#pragma unroll
  for (int i = 0; i < 8; ++i) f[i] = (f16_t)in[i];
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
  return tw_mfma32<bf16_t>(w, x, acc);
}
template <>
__device__ __forceinline__ f32x4_t sk_mfma<f16_t>(const u32x4_t& w, const u32x4_t& x, f32x4_t acc) {
  return tw_mfma32<f16_t>(w, x, acc);
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

typedef int v8i_t __attribute__((ext_vector_type(8)));
typedef unsigned short u16x2_t __attribute__((ext_vector_type(2)));
typedef short s16x2_t __attribute__((ext_vector_type(2)));

// value of lane ^ 16 / lane ^ 32 (bit pattern; after the swap {a, b} = {own, partner} whichever half the lane is in)
__device__ __forceinline__ unsigned sk_from_xor16(unsigned v) {
  float a = __builtin_bit_cast(float, v), b = a;
  tw_swap16(a, b);
  return __builtin_bit_cast(unsigned, a) ^ __builtin_bit_cast(unsigned, b) ^ v;
}
__device__ __forceinline__ unsigned sk_from_xor32(unsigned v) {
  float a = __builtin_bit_cast(float, v), b = a;
  tw_swap32(a, b);
  return __builtin_bit_cast(unsigned, a) ^ __builtin_bit_cast(unsigned, b) ^ v;
}

// MXFP8 quantisation of one 128-k operand step held by a wavefront: lane (kq = lane>>4, idx = lane&15) holds 32 bf16
// values, 4 fragments of 8; byte 8m + e of its 32-B operand <- fragment m, element e.
// Block structure = what v_mfma_scale_f32_16x16x128_f8f6f4 scales together (measured with tools/dbg/mx_probe2.hip: the
// instruction's true k of operand byte q in lane group kq is (q/16)*64 + kq*16 + q%16, and the scale of the 32-k block t
// is byte 0 of the scale register in lane t*16 + idx): block (u, h) = 16-byte half h of the two lane groups kq = 2u, 2u+1,
// its scale lives in lane group h*2 + u.  Hence: per-half maxima, one exchange with lane^16, convert, and one exchange
// with lane^32 to put each scale byte where the hardware reads it.
// Scale byte sb = max(E - 7, 1), E = biased exponent of the block's largest magnitude; elements = RNE_e4m3(v / 2^(sb-127))
// (|.| < 256, never overflows).  One exponent more conservative than the OCP MX convention (E - 8 with saturation)
// because the gfx950 converts return NaN instead of saturating above 464.  Must be called by all 64 lanes.
__device__ __forceinline__ v8i_t sk_quant_mx8(const u32x4_t (&xv)[4], int& scale_operand) {
  u16x2_t mx[2] = {{0, 0}, {0, 0}};
#pragma unroll
  for (int m = 0; m < 4; ++m)
#pragma unroll
    for (int q = 0; q < 4; ++q) mx[m >> 1] = __builtin_elementwise_max(mx[m >> 1], __builtin_bit_cast(u16x2_t, xv[m][q] & 0x7fff7fffu));
  const unsigned own = max((unsigned)mx[0][0], (unsigned)mx[0][1]) | (max((unsigned)mx[1][0], (unsigned)mx[1][1]) << 16);
  const unsigned oth = sk_from_xor16(own);
  const int sb0 = max((int)(max(own & 0xffffu, oth & 0xffffu) >> 7) - 7, 1);
  const int sb1 = max((int)(max(own >> 16, oth >> 16) >> 7) - 7, 1);
  const float X0 = __builtin_bit_cast(float, (unsigned)sb0 << 23), X1 = __builtin_bit_cast(float, (unsigned)sb1 << 23);  // 2^(sb-127)
  v8i_t out;
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    // named pairs of the whole vector, not bit_cast<bf16x2>(xv[m][q]): hipcc (ROCm 7.2) folds the latter to element 0 (NB at the top)
    const bf16x8_t a = __builtin_bit_cast(bf16x8_t, xv[m]);
    const float X = m < 2 ? X0 : X1;
    s16x2_t lo = {0, 0}, hi = {0, 0};
    lo = __builtin_amdgcn_cvt_scalef32_pk_fp8_bf16(lo, __builtin_shufflevector(a, a, 0, 1), X, false);
    lo = __builtin_amdgcn_cvt_scalef32_pk_fp8_bf16(lo, __builtin_shufflevector(a, a, 2, 3), X, true);
    hi = __builtin_amdgcn_cvt_scalef32_pk_fp8_bf16(hi, __builtin_shufflevector(a, a, 4, 5), X, false);
    hi = __builtin_amdgcn_cvt_scalef32_pk_fp8_bf16(hi, __builtin_shufflevector(a, a, 6, 7), X, true);
    out[2 * m] = __builtin_bit_cast(int, lo);
    out[2 * m + 1] = __builtin_bit_cast(int, hi);
  }
  // scale of block t belongs in lane group t: t = 0 <- (u 0, h 0), 1 <- (u 1, h 0), 2 <- (u 0, h 1), 3 <- (u 1, h 1); this
  // lane knows (u = kq/2, h = 0 and 1), the lane group kq^2 knows the other u
  const unsigned mine = (unsigned)sb0 | ((unsigned)sb1 << 8);
  const unsigned far = sk_from_xor32(mine);
  const int kq = (threadIdx.x >> 4) & 3;
  const unsigned pick = (kq == 0) ? (mine & 0xffu) : (kq == 3) ? (mine >> 8) : (kq == 1) ? (far & 0xffu) : (far >> 8);
  scale_operand = (int)(pick & 0xffu);
  return out;
}

// 16 e4m3 bytes -> two bf16 MFMA operands (bytes 0..7, 8..15), each value multiplied by X = 2^(scale byte - 127): exact in bf16
// (3 mantissa bits times a power of two).  v_cvt_scalef32_pk_bf16_fp8 converts one byte pair per instruction.
__device__ __forceinline__ void sk_widen8(const u32x4_t& raw, float X, u32x4_t& lo, u32x4_t& hi) {
  lo[0] = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8((int)raw[0], X, false));
  lo[1] = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8((int)raw[0], X, true));
  lo[2] = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8((int)raw[1], X, false));
  lo[3] = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8((int)raw[1], X, true));
  hi[0] = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8((int)raw[2], X, false));
  hi[1] = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8((int)raw[2], X, true));
  hi[2] = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8((int)raw[3], X, false));
  hi[3] = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8((int)raw[3], X, true));
}

// (sum, sum of squares) of the 16 bytes of one activation fragment, accumulated per lane
template <typename T> __device__ __forceinline__ void sk_stats(const u32x4_t& v, float& s, float& ss);
template <> __device__ __forceinline__ void sk_stats<bf16_t>(const u32x4_t& v, float& s, float& ss) {
  const bf16x8_t a = __builtin_bit_cast(bf16x8_t, v);
  const bf16x2_t one = {(bf16_t)1.0f, (bf16_t)1.0f};
  // named pairs, not a loop over v[i]: see the NB at the top of this file
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
template <> __device__ __forceinline__ void sk_stats<f16_t>(const u32x4_t& v, float& s, float& ss) {
  const f16x8_t a = __builtin_bit_cast(f16x8_t, v);
  const f16x2_t one = {(f16_t)1.0f, (f16_t)1.0f};
  const f16x2_t p0 = __builtin_shufflevector(a, a, 0, 1), p1 = __builtin_shufflevector(a, a, 2, 3);
  const f16x2_t p2 = __builtin_shufflevector(a, a, 4, 5), p3 = __builtin_shufflevector(a, a, 6, 7);
  s = tw_dot2<f16_t>(p0, one, s);
  ss = tw_dot2<f16_t>(p0, p0, ss);
  s = tw_dot2<f16_t>(p1, one, s);
  ss = tw_dot2<f16_t>(p1, p1, ss);
  s = tw_dot2<f16_t>(p2, one, s);
  ss = tw_dot2<f16_t>(p2, p2, ss);
  s = tw_dot2<f16_t>(p3, one, s);
  ss = tw_dot2<f16_t>(p3, p3, ss);
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
//
// W8 = MXFP8 weights (TW_BF16_MXFP8 contexts): the weight operand is OCP e4m3 with one power-of-two scale per 32 values
// (quant_mx8_kernel below), the bf16 activation fragments are quantised the same way in registers (sk_quant_mx8: a scale
// block is the same 16-byte half of two neighbouring lane groups, so one lane exchange finds its maximum and a second one
// puts the scale byte where the instruction reads it), and the product runs on v_mfma_scale_f32_16x16x128_f8f6f4, which
// applies both block scales in hardware.  One step is then 128 k: 2 KiB of weights + 64 scale bytes per tile, half the
// bytes of the bf16 kernel.
// CG = groups of 16 streams (1, 2 or 4: up to 64 streams per launch).  Every weight fragment is used for all groups, so the
// weight stream - the dominant cost - is paid once per launch whatever the number of streams; activations, accumulators
// and the epilogue are per group (group g of a fragment-major activation buffer starts at element g*16*K).
// TR = weight rows per workgroup tile (16, or 8 / 4 for the narrow projections): with N = 1280 a 16-row tiling yields only 80
// workgroups on a 256-CU chip and every CU has to take in 41 KB (K = 1280) or 164 KB (K = 5120) at ~50 GB/s; TR = 8 (4)
// spreads the same bytes over 160 (320) CUs.  The MFMA still contracts a 16-row A operand whose rows >= TR are zero (lanes
// fr >= TR issue no request): matrix-core time is not what bounds these launches.  Weight layout for TR < 16:
// [tile][step][kq * TR + fr][E] (tile_weights_kernel), i.e. a wavefront request is TR * 64 contiguous bytes.
// A16 (with W8) = "W8A16": the MXFP8 weight fragments are widened to bf16 in registers (sk_widen8: value x block scale, exact)
// and contracted with the UNQUANTISED bf16 activations on v_mfma_f32_16x16x32_bf16 - the same bytes from HBM as W8, no
// activation quantisation (neither its error nor its ~0.3 us of lane exchanges on the critical path of a latency-bound launch).
// A weight lane needs the scales of ITS two 32-value blocks (16-byte half h of lane groups 2u, 2u+1, u = kq / 2); the scale array
// holds block t = 2h + u in lane group t (where the scaled MFMA reads it), so the lane fetches groups u and 2 + u: two byte loads
// per step instead of one, no exchange.  Several groups of 16 streams (CG > 1) are available in this mode only.
//
// ONE reduction order for every number of rows (round 6).  A stream's result must not depend on who else is in the launch - the
// serving hub mixes sessions freely, and the draft-and-verify prefill (api.hip: verify_core) is only exact if a position computed in a
// 64-row launch has the bits the one-row step would have produced.  Every kernel flavour contracts a given k step with the same
// instruction and walks a wavefront's K slice in ascending order, so results can only differ through HOW K IS SPLIT over wavefronts.
// All projections use 8 slices [w S / 8, (w + 1) S / 8) summed in ascending order - except the long-K residual projection (fc2,
// K >= 4096), which takes 16 wavefronts when the launch has one group of streams (half the steps per wavefront on the critical
// path) and 8 with several groups (registers).  Its canonical order is therefore defined over SIXTEEN slices v = 0..15,
// P_v = sum over steps [v S / 16, (v + 1) S / 16), result = sum_{w = 0..7} (P_2w + P_2w+1): the 16-wavefront kernel adds its LDS tiles
// in pairs, the 8-wavefront kernel (SPLIT2) keeps two accumulators per group - the two halves of its slice - and adds them in
// registers before the tile goes to LDS.  Bit-identical by construction; tests/test_gpu_parity.py::test_rows_of_a_launch_do_not_matter.
template <typename T, int NW, int SK_MAXS, bool LN, int EPI, bool MULTI, bool W8, int CG, int TR, bool A16 = false, int MODE = 0, bool SPLIT2 = false>
__global__ __launch_bounds__(NW * 64, (CG > 1 ? ((NW >= 16 || (TW_CG_RING && !MULTI && !W8 && MODE == 2)) ? 4 : 2) : (NW >= 16 ? 4 : (W8 ? (SK_MAXS <= 2 ? 4 : 2) : (SK_MAXS <= 5 ? 4 : 2)))))
void skinny_mfma_kernel(const void* x_arg, const void* w_arg, int k_arg, int b_arg, int n_arg, int rg_arg,
                        const unsigned char* wscale_arg, const void* bias_arg, const void* res_arg, const float* gw_arg, GemvArgs a) {
  // The first arguments repeat what the request addresses are formed from (operands, K, B, N, tiles per workgroup): gfx950
  // delivers the leading kernel-argument dwords in SGPRs with the wave (kernarg preload, build.py: -amdgpu-kernarg-preload-count),
  // so the activation and weight requests leave without waiting for a scalar load; everything else (epilogue operands, outputs)
  // stays in the GemvArgs block behind them - a struct is never preloaded - and arrives while those requests are in flight.
  static_assert(!W8 || ElemTraits<T>::kCode == 1, "MXFP8 weights go with bf16 activations");
  static_assert(TR == 16 || (!MULTI && EPI != SK_KV && EPI != SK_F32), "narrow tiles: plain / residual / GELU projections only");
  static_assert(!W8 || TR == 16 || TR == 8, "MXFP8 weights: 16- or 8-row tiles");
  static_assert(!A16 || W8, "A16 is a flavour of the MXFP8-weight kernel");
  static_assert(!W8 || CG == 1 || A16, "several stream groups with MXFP8 weights: W8A16 only");
  static_assert(!SPLIT2 || (NW == 8 && !LN && !MULTI && CG > 1), "SPLIT2: the 8-wavefront flavour of the long-K residual projection");
  constexpr int E = ElemTraits<T>::kPer16B;
  constexpr int XPS = W8 ? 4 : 1;  // 32-wide activation fragments per MFMA step
  constexpr int WPS = W8 ? 2 : 1;  // 16-B weight requests per MFMA step
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  __shared__ float pstat[NW][CG * 16][2];  // per wavefront and stream: (sum, sum of squares) over the wavefront's K slice
  const T* x = reinterpret_cast<const T*>(x_arg);
  const T* W = reinterpret_cast<const T*>(w_arg);
  const unsigned char* wscale = wscale_arg;
  const T* bias = reinterpret_cast<const T*>(bias_arg);
  const T* res = reinterpret_cast<const T*>(res_arg);
  const float* gw_p = gw_arg;
  const float* cb_p = a.ln_cb;
  const int K = k_arg, N = n_arg, B = b_arg, ldy = a.ldy, RG = rg_arg, d_model = a.d_model;
  const long long cache_bstride = a.cache_bstride, cache_hstride = a.cache_hstride;
  T* y = reinterpret_cast<T*>(a.y);
  float* y_f32 = a.y_f32;
  T* kcache = reinterpret_cast<T*>(a.kcache);
  T* vcache = reinterpret_cast<T*>(a.vcache);
  const DecState* stt = a.stt;
  float* u_p = a.u;              // "cross query ahead" (tw_common.h): float32 [B][d_model] pre-activation, null = off
  float* stats_p = a.stats;
  const int nsplit = a.nsplit;   // SK_RES: rows >= nsplit accumulate into u (the launcher sets N when there is no such half)
  const int rows_streams = a.rows_streams;   // SK_KV: rows mode of the cache scatter (tw_row_of)
  // (1) what the activation / weight requests need: delivered with the wave (leading arguments), nothing to wait for
  asm volatile("" ::"s"(x), "s"(W), "s"(K), "s"(N), "s"(B), "s"(RG), "s"(wscale));
  int cur_pos = 0;
#ifdef TW_PROBE_TS
  unsigned long long probe_t0 = 0;
#endif
  TW_TS(0);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 15, kq = lane >> 4;
  const int S = K / (4 * E * XPS);  // MFMA steps per row (32 k each; 128 k with MXFP8 weights)
  const int s_lo = (int)((long long)wave * S / NW), s_hi = (int)((long long)(wave + 1) * S / NW);
  float* red = reinterpret_cast<float*>(smem);  // [NW][CG][256]
  const int n_tiles = (N + TR - 1) / TR;
  const int tile0 = blockIdx.x * RG;
  const int ej = (tid >> 4) & 15, ei = tid & 15;  // epilogue role of threads 0..255: stream ej, tile row ei (rows >= TR idle)

  // MODE (several groups of streams only; picked by the launcher): 0 = rounds of SK_MAXS fragments, 2 = operand rings over the
  // wavefront's whole K slice (TW_CG_RING above; SK_MAXS = the slice's step count).  (1 - rounds with the next round's weights
  // requested a round ahead - was measured in round 5 and dropped: fc2 at 64 streams 12.1 -> 12.6 us, profiles/r05_projection_probe.txt)
  constexpr bool RING = CG > 1 && TW_CG_RING && !W8 && MODE == 2;
  // RING with MULTI = DUAL (round 5's "two tiles per workgroup", back in round 6): the workgroup contracts TWO weight tiles (tile0,
  // tile0 + 1; the launcher sets RG = 2) against ONE sweep of the activation rings - half the activation (L2) traffic of two workgroups,
  // 160 instead of 320 workgroups for the 5120-row launches (QKV, fc1).  Round 5 removed it because stream 63 of 64 decoded other ids
  // than stream 0 on the same audio; the cause was not in this structure but in the LayerNorm variance, which hipcc fused differently
  // for the two stream groups an epilogue thread finishes (profiles/r06_mode3_root_cause.txt) - spelled out now (tw_ln_scalars).
  constexpr bool DUAL = RING && MULTI;
  constexpr int DW = RING ? (SK_MAXS < 5 ? SK_MAXS : 5) : 1;   // ring depths (steps)
  constexpr int DX = RING ? (SK_MAXS < 3 ? SK_MAXS : 3) : 1;
  u32x4_t rw[DW], rx[DX][CG];
  u32x4_t rw2[DUAL ? DW : 1];
  // several groups: the two 256-thread halves of the workgroup take alternate groups in the epilogue (TW_CG_EPI_ALL)
  constexpr bool EALL = CG > 1 && TW_CG_EPI_ALL && NW >= 8;
  constexpr int GPT = EALL ? CG / 2 : CG;     // groups per epilogue thread
  u32x4_t wq[SK_MAXS * WPS], xq[CG][SK_MAXS * XPS];
  int wsc[SK_MAXS];  // MXFP8: scale byte of the weight block this lane feeds to the scaled MFMA (A16: of its own first block)
  int wsc2[A16 ? SK_MAXS : 1];  // A16: scale byte of its own second block
  const int eh = EALL ? ((tid >> 8) & 1) : 0;   // which half of the workgroup this epilogue thread belongs to
  float e_c = 0.f, e_gw = 0.f, e_res[GPT];
#pragma unroll
  for (int g = 0; g < GPT; ++g) e_res[g] = 0.f;
  // --- request helpers: all unconditional, addresses clamped into the matrix ---
  auto load_epi = [&](int tile, float& c, float& gwv, float (&r)[GPT]) {
    const int n = min(tile * TR + min(ei, TR - 1), N - 1);
    const bool second = EPI == SK_RES && n >= nsplit;   // row of the composed half: its "residual" is u, it has no bias
    if (LN) {
      gwv = gw_p[n];
      c = cb_p[n];
    } else {
      const float v = (float)(bias ? bias : W)[(bias && !second) ? n : 0];
      c = (bias && !second) ? v : 0.f;
    }
    if (EPI == SK_RES) {
      // both operands are requested unconditionally (clamped addresses) and selected afterwards: no load inside a branch
      const float* up = u_p ? u_p : reinterpret_cast<const float*>(W);
      const int usz = u_p ? N - nsplit : 0;
      const int nu = (u_p && second) ? n - nsplit : 0;
#pragma unroll
      for (int gi = 0; gi < GPT; ++gi) {
        const int g = EALL ? gi * 2 + eh : gi;
        const float rb = (float)res[(long long)g * 16 * nsplit + tw_xt_index<T>(ej, second ? 0 : n)];
        const float ru = up[(long long)min(g * 16 + ej, B - 1) * usz + nu];
        r[gi] = second ? ru : rb;
      }
    }
  };
  const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<T*>(W), 0, TR == 16 ? 0 : (int)((long long)n_tiles * S * 4 * TR * 16 * (W8 ? 2 : 1)), 0x00020000);
  const __amdgpu_buffer_rsrc_t wsr = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<unsigned char*>(wscale), 0, (W8 && TR != 16) ? (int)((long long)n_tiles * S * 4 * TR) : 0, 0x00020000);
  auto load_w_into = [&](auto& wq, auto& wsc, auto& wsc2, int tile, int s0) {  // fragment-major weights: 1 KiB contiguous per wavefront request
    const int tl = min(tile, n_tiles - 1);
    if (W8 && TR == 16) {
      const unsigned char* wt = reinterpret_cast<const unsigned char*>(W) + ((long long)tl * S * 128 + lane) * 16;
      const unsigned char* ws = wscale + (long long)tl * S * 64 + lane;
#pragma unroll
      for (int i = 0; i < SK_MAXS; ++i) {
        const long long st = min(s0 + i, S - 1);
        wq[2 * i] = sk_load_w<unsigned char>(wt + st * 2048);
        wq[2 * i + 1] = sk_load_w<unsigned char>(wt + st * 2048 + 1024);
        if constexpr (A16) {
          wsc[i] = wscale[(long long)tl * S * 64 + st * 64 + (kq >> 1) * 16 + fr];
          wsc2[i] = wscale[(long long)tl * S * 64 + st * 64 + (2 + (kq >> 1)) * 16 + fr];
        } else {
          wsc[i] = ws[st * 64];
        }
      }
    } else if (W8) {
      // 8-row MXFP8 tiles (quant_mx8_kernel with tr = 8): per (tile, 128-k step) [half][kq*8 + row][16 B] + 32 scale bytes;
      // the lanes of rows >= 8 are out of range of the descriptors (no request, zero data, zero scale)
      const unsigned lane_off = (fr < TR) ? (unsigned)((kq * TR + fr) * 16) : 0x80000000u;
      const unsigned sc_off = (fr < TR) ? (unsigned)(kq * TR + fr) : 0x80000000u;
      const unsigned tile_off = (unsigned)tl * (unsigned)(S * 8 * TR * 16);   // S steps x 2 halves x 4*TR lanes x 16 B
#pragma unroll
      for (int i = 0; i < SK_MAXS; ++i) {
        const unsigned st = (unsigned)min(s0 + i, S - 1);
        const unsigned o = lane_off + tile_off + st * (unsigned)(8 * TR * 16);
        wq[2 * i] = __builtin_bit_cast(u32x4_t, __builtin_amdgcn_raw_buffer_load_b128(wr, o, 0, 2 /* nt */));
        wq[2 * i + 1] = __builtin_bit_cast(u32x4_t, __builtin_amdgcn_raw_buffer_load_b128(wr, o + (unsigned)(4 * TR * 16), 0, 2));
        if constexpr (A16) {
          const unsigned sc_a = (fr < TR) ? (unsigned)((kq >> 1) * TR + fr) : 0x80000000u;
          const unsigned sc_b = (fr < TR) ? (unsigned)((2 + (kq >> 1)) * TR + fr) : 0x80000000u;
          wsc[i] = (int)__builtin_amdgcn_raw_buffer_load_b8(wsr, sc_a + ((unsigned)tl * S + st) * (unsigned)(4 * TR), 0, 0);
          wsc2[i] = (int)__builtin_amdgcn_raw_buffer_load_b8(wsr, sc_b + ((unsigned)tl * S + st) * (unsigned)(4 * TR), 0, 0);
        } else {
          wsc[i] = (int)__builtin_amdgcn_raw_buffer_load_b8(wsr, sc_off + ((unsigned)tl * S + st) * (unsigned)(4 * TR), 0, 0);
        }
      }
    } else if (TR == 16) {
      const T* wt = W + ((long long)tl * S * 64 + lane) * E;
#pragma unroll
      for (int i = 0; i < SK_MAXS; ++i) wq[i] = sk_load_w<T>(wt + (long long)min(s0 + i, S - 1) * (64 * E));
    } else {
      // narrow tile: only the lanes of weight rows fr < TR request (through a buffer descriptor: the others are out of range
      // and return zeros without a memory transaction)
      const unsigned lane_off = (fr < TR) ? (unsigned)((kq * TR + fr) * 16) : 0x80000000u;
      const unsigned tile_off = (unsigned)tl * (unsigned)(S * 4 * TR * 16);
#pragma unroll
      for (int i = 0; i < SK_MAXS; ++i)
        wq[i] = __builtin_bit_cast(u32x4_t, __builtin_amdgcn_raw_buffer_load_b128(
            wr, lane_off + tile_off + (unsigned)(min(s0 + i, S - 1) * (4 * TR * 16)), 0, 2 /* nt */));
    }
  };
  auto load_w = [&](int tile, int s0) { load_w_into(wq, wsc, wsc2, tile, s0); };
  // fragment-major activations: lane (stream fr, k-group kq) of each 32-k step.  Requested through a buffer descriptor so
  // that the lanes of streams >= B are out of range: they return zeros WITHOUT a memory request, i.e. a launch for one
  // stream moves 1/16 of the activation bytes of a launch for 16 (the block is re-read by every workgroup: at B = 1 that
  // is the difference between 160 KB and 10 KB per workgroup for fc2).
  const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(x), 0, CG * 16 * K * (int)sizeof(T), 0x00020000);
  unsigned xoff[CG];
#pragma unroll
  for (int g = 0; g < CG; ++g)
    xoff[g] = (g * 16 + fr < B) ? (unsigned)((g * 16 * K + lane * E) * (int)sizeof(T)) : 0x80000000u;
  auto load_x = [&](int s0) {
    if constexpr (CG > 1 && TW_CG_ORDER) {   // step-major: the order mfma_round consumes them in
#pragma unroll
      for (int i = 0; i < SK_MAXS; ++i)
#pragma unroll
        for (int g = 0; g < CG; ++g)
#pragma unroll
          for (int m = 0; m < XPS; ++m) {
            const unsigned step_bytes = (unsigned)((min(s0 + i, S - 1) * XPS + m) * (64 * E) * (int)sizeof(T));
            xq[g][i * XPS + m] = __builtin_bit_cast(u32x4_t, __builtin_amdgcn_raw_buffer_load_b128(xr, xoff[g] + step_bytes, 0, 0));
          }
    } else {
#pragma unroll
      for (int g = 0; g < CG; ++g)
#pragma unroll
        for (int i = 0; i < SK_MAXS; ++i)
#pragma unroll
          for (int m = 0; m < XPS; ++m) {
            const unsigned step_bytes = (unsigned)((min(s0 + i, S - 1) * XPS + m) * (64 * E) * (int)sizeof(T));
            xq[g][i * XPS + m] = __builtin_bit_cast(u32x4_t, __builtin_amdgcn_raw_buffer_load_b128(xr, xoff[g] + step_bytes, 0, 0));
          }
    }
  };

  auto ring_w = [&](int step, int tile_ofs = 0) -> u32x4_t {   // one weight fragment of this workgroup's tile (RING: every step is inside the matrix)
    const int tl = min(tile0 + tile_ofs, n_tiles - 1);
    if constexpr (TR == 16) {
      return sk_load_w<T>(W + ((long long)tl * S * 64 + lane) * E + (long long)step * (64 * E));
    } else {
      const unsigned lane_off = (fr < TR) ? (unsigned)((kq * TR + fr) * 16) : 0x80000000u;
      return __builtin_bit_cast(u32x4_t, __builtin_amdgcn_raw_buffer_load_b128(
          wr, lane_off + (unsigned)tl * (unsigned)(S * 4 * TR * 16) + (unsigned)(step * (4 * TR * 16)), 0, 2 /* nt */));
    }
  };
  auto ring_x = [&](int g, int step) -> u32x4_t {
    return __builtin_bit_cast(u32x4_t, __builtin_amdgcn_raw_buffer_load_b128(xr, xoff[g] + (unsigned)(step * (64 * E) * (int)sizeof(T)), 0, 0));
  };
  if constexpr (RING) {
    // HBM first, then the first DX steps of activations in consumption order; everything else is requested as registers free up
#pragma unroll
    for (int i = 0; i < DW; ++i) {
      rw[i] = ring_w(s_lo + i);
      if constexpr (DUAL) rw2[i] = ring_w(s_lo + i, 1);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < DX; ++i)
#pragma unroll
      for (int g = 0; g < CG; ++g) rx[i][g] = ring_x(g, s_lo + i);
    __builtin_amdgcn_sched_barrier(0);
  } else if constexpr (CG > 1 && TW_CG_ORDER) {
    // several groups of streams: a wavefront asks for up to 20 KiB of activations per round.  The HBM requests go first (their
    // latency is the long one; vmcnt retires in order, so by the time an activation fragment has arrived its weights have too)
    load_w(tile0, s_lo);
    __builtin_amdgcn_sched_barrier(0);
    load_x(s_lo);
    __builtin_amdgcn_sched_barrier(0);
  } else {
    load_x(s_lo);
    __builtin_amdgcn_sched_barrier(0);  // request order = consumption order: hipcc must not reorder the groups
    load_w(tile0, s_lo);
    __builtin_amdgcn_sched_barrier(0);
  }
  // (2) everything else (epilogue operands, outputs, cache geometry): ONE batch of scalar loads from the argument block, waited for
  // here, behind the operand requests that are already on their way
  asm volatile("" ::"s"(bias), "s"(res), "s"(gw_p), "s"(cb_p), "s"(a.gelu), "s"(ldy), "s"(d_model), "s"(cache_bstride), "s"(cache_hstride), "s"(y),
               "s"(y_f32), "s"(kcache), "s"(vcache), "s"(stt), "s"(u_p), "s"(nsplit), "s"(stats_p), "s"(rows_streams));
#ifdef TW_PROBE_TS
  cur_pos = stt->pos;
#else
  if (EPI == SK_KV) cur_pos = stt->pos;
#endif
  load_epi(tile0, e_c, e_gw, e_res);
  __builtin_amdgcn_sched_barrier(0);  // ... nor hoist arithmetic between the requests
  TW_TS(1);

  float mean[GPT], rstd[GPT];
#pragma unroll
  for (int g = 0; g < GPT; ++g) { mean[g] = 0.f; rstd[g] = 1.f; }
  const int n_grp = MULTI ? RG : 1;
  f32x4_t accd[DUAL ? CG : 1];   // DUAL: the second tile's accumulators, filled together with the first tile's
  for (int grp = 0; grp < n_grp; ++grp) {
    const int tile = tile0 + grp;
    f32x4_t acc[CG], acc2[SPLIT2 ? CG : 1];   // acc2: the second half of the wavefront's K slice (SPLIT2, see the kernel's header)
    float ps[CG], pss[CG];
#pragma unroll
    for (int g = 0; g < CG; ++g) { acc[g] = f32x4_t{0.f, 0.f, 0.f, 0.f}; ps[g] = 0.f; pss[g] = 0.f; }
#pragma unroll
    for (int g = 0; g < (SPLIT2 ? CG : 1); ++g) acc2[g] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    const int s_mid = SPLIT2 ? (int)((long long)(2 * wave + 1) * S / (2 * NW)) : 0x7fffffff;   // first step of slice 2 wave + 1 of 16
    auto mfma_round = [&](int s0) {
#pragma unroll
      for (int i = 0; i < SK_MAXS; ++i) {
        const bool on = s0 + i < s_hi;  // wave-uniform: steps past this wavefront's K slice contribute zero
        const bool second = SPLIT2 && s0 + i >= s_mid;   // wave-uniform: which of the two accumulator sets this step belongs to
        const u32x4_t zero = u32x4_t{0u, 0u, 0u, 0u};
        if constexpr (A16) {
          // fragments m = 0, 1 live in the first 16 bytes (block h = 0), m = 2, 3 in the second (h = 1)
          u32x4_t wf[4];
          sk_widen8(on ? wq[2 * i] : zero, __builtin_bit_cast(float, (unsigned)(wsc[i] & 0xff) << 23), wf[0], wf[1]);
          sk_widen8(on ? wq[2 * i + 1] : zero, __builtin_bit_cast(float, (unsigned)(wsc2[i] & 0xff) << 23), wf[2], wf[3]);
          if (SPLIT2 && second) {   // (a branch around matrix instructions only: no load inside)
#pragma unroll
            for (int g = 0; g < (SPLIT2 ? CG : 1); ++g)
#pragma unroll
              for (int m = 0; m < 4; ++m) acc2[g] = sk_mfma<T>(wf[m], on ? xq[g][i * 4 + m] : zero, acc2[g]);
          } else {
#pragma unroll
          for (int g = 0; g < CG; ++g)
#pragma unroll
            for (int m = 0; m < 4; ++m) {
              const u32x4_t xv = on ? xq[g][i * 4 + m] : zero;
              if (LN && (!MULTI || grp == 0)) sk_stats<T>(xv, ps[g], pss[g]);
              acc[g] = sk_mfma<T>(wf[m], xv, acc[g]);
            }
          }
        } else if (W8) {
          const u32x4_t w0 = on ? wq[2 * i] : zero, w1 = on ? wq[2 * i + 1] : zero;
          const v8i_t wb = {(int)w0[0], (int)w0[1], (int)w0[2], (int)w0[3], (int)w1[0], (int)w1[1], (int)w1[2], (int)w1[3]};
#pragma unroll
          for (int g = 0; g < CG; ++g) {
            u32x4_t xv[4];
#pragma unroll
            for (int m = 0; m < 4; ++m) {
              xv[m] = on ? xq[g][i * 4 + m] : zero;
              if (LN && (!MULTI || grp == 0)) sk_stats<T>(xv[m], ps[g], pss[g]);
            }
            int xs;
            const v8i_t xb = sk_quant_mx8(xv, xs);
            acc[g] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(wb, xb, acc[g], 0, 0, 0, wsc[i], 0, xs);
          }
        } else {
          const u32x4_t wv = on ? wq[i] : zero;
          if (SPLIT2 && second) {
#pragma unroll
            for (int g = 0; g < (SPLIT2 ? CG : 1); ++g) acc2[g] = sk_mfma<T>(wv, on ? xq[g][i] : zero, acc2[g]);
          } else {
#pragma unroll
          for (int g = 0; g < CG; ++g) {
            const u32x4_t xv = on ? xq[g][i] : zero;
            if (LN && (!MULTI || grp == 0)) sk_stats<T>(xv, ps[g], pss[g]);
            acc[g] = sk_mfma<T>(wv, xv, acc[g]);
          }
          }
        }
      }
    };
    if constexpr (RING) {
      static_assert(!SPLIT2 || SK_MAXS % 2 == 0, "SPLIT2 rings: the slice's midpoint is a step boundary");
      static_assert(!(SPLIT2 && DUAL), "SPLIT2 is the long-K residual projection, DUAL the 5120-row launches");
      if (!DUAL || grp == 0) {
#pragma unroll
        for (int g = 0; g < (DUAL ? CG : 1); ++g) accd[g] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < SK_MAXS; ++i) {
#pragma unroll
          for (int g = 0; g < CG; ++g) {
            const u32x4_t xv = rx[i % DX][g];
            if (LN) sk_stats<T>(xv, ps[g], pss[g]);
            if constexpr (SPLIT2) {   // (compile-time: S = NW * SK_MAXS, so s_mid = s_lo + SK_MAXS / 2)
              if (i >= SK_MAXS / 2) acc2[g] = sk_mfma<T>(rw[i % DW], xv, acc2[g]);
              else acc[g] = sk_mfma<T>(rw[i % DW], xv, acc[g]);
            } else {
              acc[g] = sk_mfma<T>(rw[i % DW], xv, acc[g]);
              if constexpr (DUAL) accd[g] = sk_mfma<T>(rw2[i % DW], xv, accd[g]);
            }
          }
          if (i + DX < SK_MAXS) {
#pragma unroll
            for (int g = 0; g < CG; ++g) rx[i % DX][g] = ring_x(g, s_lo + i + DX);
          }
          if (i + DW < SK_MAXS) {
            rw[i % DW] = ring_w(s_lo + i + DW);
            if constexpr (DUAL) rw2[i % DW] = ring_w(s_lo + i + DW, 1);
          }
          __builtin_amdgcn_sched_barrier(0);   // keep the software pipeline in this order
        }
      } else {   // DUAL, second tile: contracted together with the first (same K slices, same order: bit-identical to a tile of its own)
#pragma unroll
        for (int g = 0; g < (DUAL ? CG : 1); ++g) acc[g] = accd[g];
      }
    } else {
    mfma_round(s_lo);  // operands already in flight
    for (int s0 = s_lo + SK_MAXS; s0 < s_hi; s0 += SK_MAXS) {  // K longer than one round of fragments
      if constexpr (CG > 1 && TW_CG_ORDER) {
        load_w(tile, s0);
        __builtin_amdgcn_sched_barrier(0);
        load_x(s0);
      } else {
        load_x(s0);
        load_w(tile, s0);
      }
      mfma_round(s0);
    }
    }
    // operands of the tile after this one are requested now, behind the reduction and the epilogue of the current one
    // (MULTI is only launched with a single round of fragments per tile, so the activation fragments stay in registers)
    float n_c = 0.f, n_gw = 0.f, n_res[GPT];
#pragma unroll
    for (int g = 0; g < GPT; ++g) n_res[g] = 0.f;
    if (MULTI) {
      const int nt = min(tile + 1, n_tiles - 1);
      if constexpr (!RING) load_w(nt, s_lo);
      load_epi(nt, n_c, n_gw, n_res);
    }
    TW_TS(2);
    // D[i = weight row (lane>>4)*4 + reg][j = stream lane&15]
    if (MULTI && grp > 0) __syncthreads();  // previous tile's readers are done with `red`
    if constexpr (SPLIT2) {   // P_2w + P_2w+1: the addition the 16-wavefront kernel performs on its LDS tiles
#pragma unroll
      for (int g = 0; g < (SPLIT2 ? CG : 1); ++g) acc[g] += acc2[g];
    }
#pragma unroll
    for (int g = 0; g < CG; ++g)
#pragma unroll
      for (int r = 0; r < 4; ++r) red[(wave * CG + g) * kRedTile + (kq * 4 + r) * TW_RED_STRIDE + fr] = acc[g][r];
    if (LN && (!MULTI || grp == 0)) {
#pragma unroll
      for (int g = 0; g < CG; ++g) {
        const float s1 = tw_xor32_sum(tw_xor16_sum(ps[g])), s2 = tw_xor32_sum(tw_xor16_sum(pss[g]));
        if (lane < 16) { pstat[wave][g * 16 + fr][0] = s1; pstat[wave][g * 16 + fr][1] = s2; }
      }
    }
    __syncthreads();
    TW_TS(3);
    if (tid < (EALL ? 512 : 256)) {
      const int j = ej, i = ei;  // stream (within its group), row: 16 consecutive rows of one stream per 16 threads
      const int n = tile * TR + i;
#pragma unroll
      for (int gi = 0; gi < GPT; ++gi) {
        const int g = EALL ? gi * 2 + eh : gi;
        const int jg = g * 16 + j;  // stream
        float v = 0.f;
        if constexpr (NW >= 16) {   // sixteen slices added in pairs: the order the 8-wavefront SPLIT2 flavour reproduces (kernel header)
#pragma unroll
          for (int w = 0; w < NW; w += 2)
            v += red[(w * CG + g) * kRedTile + i * TW_RED_STRIDE + j] + red[((w + 1) * CG + g) * kRedTile + i * TW_RED_STRIDE + j];
        } else {
#pragma unroll
          for (int w = 0; w < NW; ++w) v += red[(w * CG + g) * kRedTile + i * TW_RED_STRIDE + j];
        }
        const float vraw = v;
        if (LN) {
          if (!MULTI || grp == 0) {
            float sx = 0.f, sxx = 0.f;
#pragma unroll
            for (int w = 0; w < NW; ++w) { sx += pstat[w][jg][0]; sxx += pstat[w][jg][1]; }
            const float inv_k = __builtin_amdgcn_rcpf((float)K);
            tw_ln_scalars(sx, sxx, inv_k, mean[gi], rstd[gi]);
          }
          v = tw_ln_apply(v, mean[gi], rstd[gi], e_gw, e_c);
        } else {
          v += e_c;
        }
        if (EPI == SK_GELU) v = gelu_exact<T>(v);
        if (EPI == SK_RES) v += e_res[gi];
        if (EPI == SK_RES && u_p) {   // (kernel-uniform) LayerNorm statistics of the residual rows of this tile, as stored
          const float xs = (tile < n_tiles && n < nsplit && jg < B && i < TR) ? (float)tw_cast<T>(v) : 0.f;
          const float s1 = tw_row16_sum(xs), s2 = tw_row16_sum(xs * xs);   // the 16 threads of one stream are one DPP row
          if (i == 0 && jg < B && tile < n_tiles && tile * TR < nsplit) {
            float* sp = stats_p + ((long long)jg * (nsplit / TR) + tile) * 2;
            sp[0] = s1;
            sp[1] = s2;
          }
        }
        if (tile < n_tiles && n < N && jg < B && i < TR) {
          if (EPI == SK_F32) {
            y_f32[(long long)jg * N + n] = v;
          } else if (EPI == SK_KV) {
            // 0: query -> y, 1: key -> K cache, 2: value -> V^T cache, 3: cross query ahead   (one predicated store; compares and a
            // host-side per-head stride instead of the 32- and 64-bit divisions by run-time values this epilogue used to carry: ~0.3 us)
            const int seg = (n >= d_model) + (n >= 2 * d_model) + (n >= 3 * d_model);
            const int nn = n - seg * d_model, hh = nn >> 6, cc = nn & 63;
            if (seg == 3) {   // cross query ahead: x . W'^T + c0 WITHOUT this LayerNorm (its own is applied by the consumer)
              u_p[(long long)jg * d_model + nn] = vraw + e_c;
            } else {
              int srow, prow;   // rows mode (prefill): row jg is stream jg % rs at position cur_pos + jg / rs
              tw_row_of(jg, rows_streams, cur_pos, srow, prow);
              const long long hb = (long long)srow * cache_bstride + (long long)hh * cache_hstride;  // (stream, head)
              T* dst = seg == 0 ? y + (long long)jg * ldy + n
                                : (seg == 1 ? kcache + hb + tw_kf_index<T>(prow, cc) : vcache + hb + tw_vtf_index<T>(prow, cc));
              *dst = tw_cast<T>(v);
            }
          } else if (EPI == SK_STORE) {
            y[(long long)jg * ldy + n] = tw_cast<T>(v);  // row-major [B][ldy] (the attention kernels' query operand)
          } else if (EPI == SK_RES) {
            if (n >= nsplit) u_p[(long long)jg * (N - nsplit) + (n - nsplit)] = v;   // composed half: u += attn . Wc^T
            else y[(long long)g * 16 * nsplit + tw_xt_index<T>(j, n)] = tw_cast<T>(v);       // residual stream, fragment-major
          } else {
            y[(long long)g * 16 * N + tw_xt_index<T>(j, n)] = tw_cast<T>(v);    // feeds the next projection: fragment-major
          }
        }
      }
    }
    if (MULTI) {
      e_c = n_c; e_gw = n_gw;
#pragma unroll
      for (int g = 0; g < GPT; ++g) e_res[g] = n_res[g];
    }
    TW_TS(4);
  }
}

// Row-major W[N][K] -> fragment-major layout read by skinny_mfma_kernel: for tile t = n/16 and step s = k/(4E) one
// 1-KiB block holds the MFMA A operand exactly as the 64 lanes consume it, lane l = kq*16 + fr <- row t*16+fr,
// 16-B vector s*4+kq.  A wavefront's K slice is then ONE contiguous run (its requests are full cache lines in address
// order, like a plain streaming copy) instead of 16 row segments of 64 B per request.  Rows >= N are zero.
template <typename T>
__global__ __launch_bounds__(256) void tile_weights_kernel(const T* __restrict__ src, T* __restrict__ dst, int N, int K, int TR) {
  constexpr int E = ElemTraits<T>::kPer16B;
  const int S = K / E / 4;
  const int LT = 4 * TR;                                            // 16-B vectors per (tile, step): 64 for 16-row tiles
  const long long v = (long long)blockIdx.x * 256 + threadIdx.x;  // destination vector index
  const long long total = (long long)((N + TR - 1) / TR) * S * LT;
  if (v >= total) return;
  const int l = (int)(v % LT);
  const long long ts = v / LT;
  const int s = (int)(ts % S);
  const int t = (int)(ts / S);
  const int n = t * TR + (l % TR), kv = s * 4 + (l / TR);
  u32x4_t val = u32x4_t{0u, 0u, 0u, 0u};
  if (n < N) val = *reinterpret_cast<const u32x4_t*>(src + (long long)n * K + (long long)kv * E);
  *reinterpret_cast<u32x4_t*>(dst + v * E) = val;
}

// Row-major bf16 W[N][K] -> MXFP8 operand of skinny_mfma_kernel<.., W8 = true>: per 16-row tile and 128-k step, 2 KiB of
// e4m3 as [half][lane][16 B] (lane = kq*16 + row%16; its 32 bytes, fragment m / element e at byte 8m + e, are
// W[row][(4s + m)*32 + kq*8 + e]: the same k set the activation lane (stream, kq) holds) followed, in a separate array, by
// one scale byte per lane (already placed in the lane the MFMA reads it from).  Quantisation = sk_quant_mx8, i.e. identical
// to what the projection kernel does to the activations.
__global__ __launch_bounds__(256) void quant_mx8_kernel(const bf16_t* __restrict__ src, unsigned char* __restrict__ dst,
                                                        unsigned char* __restrict__ scales, int N, int K, int TR) {
  const int S = K / 128;
  const long long v = (long long)blockIdx.x * 256 + threadIdx.x;  // (16-row tile, step, lane): quantisation always works on
  const long long total = (long long)((N + 15) / 16) * S * 64;    // whole wavefronts = 16 rows x 4 k-groups
  if (v >= total) return;
  const int l = (int)(v & 63);
  const long long ts = v >> 6;
  const int s = (int)(ts % S);
  const long long t = ts / S;
  const int n = (int)t * 16 + (l & 15), kq = l >> 4;
  u32x4_t xv[4];
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    xv[m] = u32x4_t{0u, 0u, 0u, 0u};
    if (n < N) xv[m] = *reinterpret_cast<const u32x4_t*>(src + (long long)n * K + (4 * s + m) * 32 + kq * 8);
  }
  int sb;  // (whole wavefronts reach this point: `total` is a multiple of 64)
  const v8i_t q = sk_quant_mx8(xv, sb);
  // storage: tiles of TR rows (16, or 8 for the narrow projections): [tile][step][half][kq * TR + row % TR][16 B]
  const int fr = l & 15;
  const long long tile = t * (16 / TR) + fr / TR;
  const int slot = kq * TR + fr % TR;
  unsigned char* blk = dst + (tile * S + s) * (long long)(8 * TR * 16);
  *reinterpret_cast<u32x4_t*>(blk + slot * 16) = u32x4_t{(unsigned)q[0], (unsigned)q[1], (unsigned)q[2], (unsigned)q[3]};
  *reinterpret_cast<u32x4_t*>(blk + 4 * TR * 16 + slot * 16) = u32x4_t{(unsigned)q[4], (unsigned)q[5], (unsigned)q[6], (unsigned)q[7]};
  scales[(tile * S + s) * (long long)(4 * TR) + slot] = (unsigned char)sb;
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
    const T wr = (T)(gk * w);
    sg += (float)wr;   // of the weights AS STORED: rstd (x W'^T - mean gw) is then exactly rstd (x - mean) W'^T
    sb += (float)beta[k] * w;
    W[(long long)n * K + k] = wr;
  }
  sg = wave_sum(sg);
  sb = wave_sum(sb);
  if (lane == 0) {
    gw[n] = sg;
    cb[n] = sb + (bias ? (float)bias[n] : 0.f);
  }
}

// Load-time composition for "cross query ahead": out = A . Bm (row-major [N][J] x [J][K], float32 accumulation, one rounding
// to T), c0 = A . bvec.  Runs once per decoder layer in tw_finalize_weights: plain 16x16 LDS tiles, nothing to tune.
template <typename T>
__global__ __launch_bounds__(256) void compose_kernel(const T* __restrict__ A, const T* __restrict__ Bm, const T* __restrict__ bvec,
                                                      T* __restrict__ out, float* __restrict__ c0, int N, int J, int K) {
  __shared__ float ta[16][17], tb[16][17];
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  const int n = blockIdx.y * 16 + ty, k = blockIdx.x * 16 + tx;
  float acc = 0.f, accb = 0.f;
  for (int j0 = 0; j0 < J; j0 += 16) {
    ta[ty][tx] = (n < N && j0 + tx < J) ? (float)A[(long long)n * J + j0 + tx] : 0.f;
    tb[ty][tx] = (j0 + ty < J && k < K) ? (float)Bm[(long long)(j0 + ty) * K + k] : 0.f;
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 16; ++j) acc = fmaf(ta[ty][j], tb[j][tx], acc);
    if (blockIdx.x == 0 && tx == 0 && bvec) {
      for (int j = 0; j < 16 && j0 + j < J; ++j) accb = fmaf(ta[ty][j], (float)bvec[j0 + j], accb);
    }
    __syncthreads();
  }
  if (n < N && k < K) out[(long long)n * K + k] = (T)acc;
  if (blockIdx.x == 0 && tx == 0 && n < N && c0) c0[n] = accb;
}

// ---------------------------------------------------------------------------------------------
// single-query attention (decoder self-attention over the growing cache, cross-attention over the
// cached encoder K/V).  One workgroup per (stream, head), NW wavefronts, each owning 64 consecutive
// keys per chunk of NW*64 keys.  Both products run on the matrix cores although there is only one
// query: K is the A operand (16 keys x 64 dims per tile) against the query broadcast over the 16 B
// columns, V^T is the A operand (16 dims x 4E keys) against the probabilities.  15/16 of the MFMA
// columns are redundant, which is free - the kernel moves 128 KiB per head (cross attention, 10 s)
// and is bound by how fast one CU can take that in, and the VALU formulation (a dot product, a
// 16-lane reduction and an exp per key and 8-byte loads) cost more issue time than the loads.
// K and V^T are stored fragment-major (tw_kf_index / tw_vtf_index in tw_common.h), written in that
// layout by the producers (cross-K/V GEMM epilogue, QKV projection epilogue): every request of a
// wavefront is 1 KiB contiguous and a wavefront's 64 keys are one 8-KiB run of each operand.
// Softmax in fp32 (scores -> LDS, block max, exp, sum); probabilities are rounded to the storage
// type for the P.V product as the reference does (HF casts attn_weights to the value dtype).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float wave_max(float v) { return tw_wave_max(v); }

// Fused query operand (FusedQ, tw_common.h): the query of the cross attention is the LayerNorm'd projection of the residual
// stream, q = rstd (u - mean gw) + cb, with u = x . W'^T accumulated by the two projections that ran before (api.hip:
// decode_core).  Every wavefront folds the partial row statistics the residual launch left behind (one coalesced 8-byte
// request per lane and 64 partials; reading the fragment-major row itself costs 64 cache lines per request and made the
// launch 3 us slower) and lane l forms element l of the head's query; the MFMA operand order is then read back from a 64-element LDS
// row PRIVATE to the wavefront (LDS operations of one wavefront stay in order: no barrier).  All requests are issued before
// the K / V^T requests of the caller and consumed after them, i.e. this arithmetic runs while the K fragments are in flight.
template <typename T>
struct QRaw {
  static constexpr int MAXP = 3;   // partial statistics per lane: n_part <= 192 (d = 1280 in 8-row tiles: 160)
  float u, gw, cb;
  f32x2_t st[MAXP];
};
template <typename T>
__device__ __forceinline__ void fq_request(QRaw<T>& r, const FusedQ& fq, int b, int h, int lane) {
  r.u = fq.u[(long long)b * fq.d + h * 64 + lane];
  r.gw = fq.gw[h * 64 + lane];
  r.cb = fq.cb[h * 64 + lane];
  const f32x2_t* sp = reinterpret_cast<const f32x2_t*>(fq.stats) + (long long)b * fq.n_part;
#pragma unroll
  for (int m = 0; m < QRaw<T>::MAXP; ++m) r.st[m] = sp[min(lane + 64 * m, fq.n_part - 1)];
}
// qrow: this wavefront's 64-element LDS row.  Returns with the element `lane` of the query stored there (type T).
template <typename T>
__device__ __forceinline__ void fq_finish(const QRaw<T>& r, const FusedQ& fq, int lane, T* qrow) {
  float s = 0.f, ss = 0.f;
#pragma unroll
  for (int m = 0; m < QRaw<T>::MAXP; ++m) {
    const bool on = lane + 64 * m < fq.n_part;
    s += on ? r.st[m][0] : 0.f;
    ss += on ? r.st[m][1] : 0.f;
  }
  s = tw_wave_sum(s);
  ss = tw_wave_sum(ss);
  const float inv_k = __builtin_amdgcn_rcpf((float)fq.d);
  float mean, rstd;
  tw_ln_scalars(s, ss, inv_k, mean, rstd);
  qrow[lane] = tw_cast<T>(tw_ln_apply(r.u, mean, rstd, r.gw, r.cb));
}

// SINGLE: n_bound <= NW*64, so every K and V^T fragment of the head is requested before anything is waited for
// (one memory round trip per head); otherwise two passes over chunks of NW*64 keys.
// n_keys = valid keys (rest masked), n_bound = keys addressable (multiple of 64, <= rows allocated per head).
// sc: LDS float[n_bound] (unnormalised probabilities on return); red: LDS float[2*NW + NW*64].
// Returns 1/L in every thread; threads 0..63 write the context vector (fragment-major activation layout).
// FQ: the query comes as a FusedQ (stream fq_b, head fq_h; qst = LDS T[NW][64]) instead of through q
template <typename T, int NW, bool SINGLE, bool FQ = false>
__device__ __forceinline__ float attn_mfma_block(const T* __restrict__ q, const T* __restrict__ kf, const T* __restrict__ vtf,
                                                 int n_keys, int n_bound, float* sc, float* red, T* __restrict__ out_base,
                                                 int out_j, int out_k0, const FusedQ& fq = FusedQ{}, int fq_b = 0, int fq_h = 0,
                                                 T* qst = nullptr) {
  constexpr int E = ElemTraits<T>::kPer16B;
  constexpr int DS = 64 / (4 * E);  // dim steps per key tile (QK^T) = key steps per 64 keys (P.V)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 15, kq = lane >> 4;
  const int n_chunks = SINGLE ? 1 : (n_bound + NW * 64 - 1) / (NW * 64);
  const int last64 = n_bound / 64 - 1;  // last addressable 64-key group
  u32x4_t qf[DS], kfr[4 * DS], vfr[4 * DS];
  auto load_k = [&](int g64) {  // 64 keys = 4 tiles x DS steps = 4*DS KiB contiguous
    const T* p = kf + ((long long)min(g64, last64) * (4 * DS) * 64 + lane) * E;
#pragma unroll
    for (int i = 0; i < 4 * DS; ++i) kfr[i] = *reinterpret_cast<const u32x4_t*>(p + (long long)i * 64 * E);
  };
  auto load_v = [&](int g64) {  // 64 keys = DS key steps x 4 dim tiles
    const T* p = vtf + ((long long)min(g64, last64) * (4 * DS) * 64 + lane) * E;
#pragma unroll
    for (int i = 0; i < 4 * DS; ++i) vfr[i] = *reinterpret_cast<const u32x4_t*>(p + (long long)i * 64 * E);
  };
  // K first: its addresses are formed from the leading (preloaded) kernel arguments alone, so these requests leave at once;
  // the query side needs the rest of the argument block (one scalar batch, pinned here) and goes out behind them
  load_k(wave);
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("" ::"s"(q), "s"(fq.u), "s"(fq.stats), "s"(fq.n_part), "s"(fq.gw), "s"(fq.cb), "s"(fq.d));
  QRaw<T> qr;
  if constexpr (FQ) {
    fq_request<T>(qr, fq, fq_b, fq_h, lane);
  } else {
#pragma unroll
    for (int ds = 0; ds < DS; ++ds) qf[ds] = *reinterpret_cast<const u32x4_t*>(q + ds * 4 * E + kq * E);
  }
  if (SINGLE) load_v(wave);
  __builtin_amdgcn_sched_barrier(0);
  if constexpr (FQ) {
    T* qrow = qst + wave * 64;
    fq_finish<T>(qr, fq, lane, qrow);
#pragma unroll
    for (int ds = 0; ds < DS; ++ds) qf[ds] = *reinterpret_cast<const u32x4_t*>(qrow + ds * 4 * E + kq * E);
  }

  // ---- scores: D[i = key (lane>>4)*4 + r][j] identical in every column j; column 0 lanes publish them ----
  float m = -1.0e30f;
  for (int c = 0; c < n_chunks; ++c) {
    const int g64 = c * NW + wave;
    if (c > 0) load_k(g64);
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      f32x4_t acc = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ds = 0; ds < DS; ++ds) acc = sk_mfma<T>(kfr[a * DS + ds], qf[ds], acc);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int t = g64 * 64 + a * 16 + kq * 4 + r;
        const float sv = (t < n_keys) ? acc[r] : -1.0e30f;
        m = fmaxf(m, sv);
        if (fr == 0 && t < n_bound) sc[t] = sv;
      }
    }
  }
  m = tw_xor32_max(tw_xor16_max(m));
  if (lane == 0) red[wave] = m;
  __syncthreads();
  float M = red[0];
#pragma unroll
  for (int w = 1; w < NW; ++w) M = fmaxf(M, red[w]);

  // ---- probabilities of this wavefront's own keys (written back to LDS for the B operand and the alignment rows) and P.V
  f32x4_t o[4];
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) o[dt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  float ls = 0.f;
  for (int c = 0; c < n_chunks; ++c) {
    const int g64 = c * NW + wave;
    if (!SINGLE) load_v(g64);
    const int t0 = g64 * 64 + lane;  // one key per lane
    float pr = 0.f;
    if (t0 < n_bound) {
      pr = (t0 < n_keys) ? expf(sc[t0] - M) : 0.f;
      sc[t0] = pr;
    }
    ls += pr;
#pragma unroll
    for (int ks = 0; ks < DS; ++ks) {
      float pv[E];
#pragma unroll
      for (int e = 0; e < E; ++e) {
        const int t = g64 * 64 + ks * 4 * E + kq * E + e;
        pv[e] = (t < n_bound) ? sc[t] : 0.f;  // same wavefront wrote these: LDS operations of a wavefront stay in order
      }
      const u32x4_t pf = pack16<T>(pv);
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) o[dt] = sk_mfma<T>(vfr[ks * 4 + dt], pf, o[dt]);
    }
  }
  ls = tw_wave_sum(ls);
  float* wl = red + NW;       // [NW]
  float* wo = red + 2 * NW;   // [NW][64]
  if (lane == 0) wl[wave] = ls;
  if (fr == 0) {
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
      for (int r = 0; r < 4; ++r) wo[wave * 64 + dt * 16 + kq * 4 + r] = o[dt][r];
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

// fp8 variant of attn_mfma_block for the cross attention of TW_BF16_MXFP8 contexts (bf16 activations): K and V^T are stored as
// e4m3 with one power-of-two scale byte per (key, head) (layouts: tw_kf8_index / tw_vtf8_index, written by gemm_epilogue_kv8),
// which halves the bytes the step has to stream - at 16 streams the cross-attention K/V are as many bytes as all weights.
// The products stay bf16 MFMAs: fragments are widened in registers (v_cvt_scalef32_pk_bf16_fp8; the K scale is applied by
// that instruction, lane = key; the V scale multiplies the probability of its key before the P.V operand is packed), so q and
// the probabilities are NOT quantised.  A 16-B fp8 vector holds 16 dims (K) / 16 keys (V^T), i.e. two bf16 MFMA operands:
// the contraction index of an MFMA is order-free, so the q / probability operands are simply gathered in the same order.
__device__ __forceinline__ void kv8_widen(const u32x4_t& raw, float X, u32x4_t& lo, u32x4_t& hi) {
  lo[0] = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8((int)raw[0], X, false));
  lo[1] = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8((int)raw[0], X, true));
  lo[2] = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8((int)raw[1], X, false));
  lo[3] = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8((int)raw[1], X, true));
  hi[0] = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8((int)raw[2], X, false));
  hi[1] = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8((int)raw[2], X, true));
  hi[2] = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8((int)raw[3], X, false));
  hi[3] = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8((int)raw[3], X, true));
}

// G = 64-key groups per wavefront (1, 2 or 3: up to 512 / 1024 / 1536 keys with 8 wavefronts, i.e. every chunk length up to
// 30 s).  At 32 registers per group and operand the whole head fits the register file, so - unlike the bf16 kernel, which
// falls back to two dependent passes above 512 keys - EVERY fragment of the head is requested before anything is waited for:
// one memory round trip per head whatever the chunk length.
template <int NW, int G, bool FQ = false>
__device__ __forceinline__ float attn_mfma_block_kv8(const bf16_t* __restrict__ q, const unsigned char* __restrict__ kf8,
                                                     const unsigned char* __restrict__ vtf8, const unsigned char* __restrict__ ksc,
                                                     const unsigned char* __restrict__ vsc, int n_keys, int n_bound, float* sc,
                                                     float* scv, float* red, bf16_t* __restrict__ out_base, int out_j, int out_k0,
                                                     const FusedQ& fq = FusedQ{}, int fq_b = 0, int fq_h = 0,
                                                     bf16_t* qst = nullptr) {
  typedef bf16_t T;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 15, kq = lane >> 4;
  const int last64 = n_bound / 64 - 1;
  u32x4_t qf[2], kraw[G][4], vraw[G][4];
  unsigned ks4[G], vsb[G];
#pragma unroll
  for (int g = 0; g < G; ++g) {  // 64 keys = 4 tiles of 1 KiB + their 64 scale bytes ([group][key % 16][tile]: one 32-bit load)
    const int gi = min(g * NW + wave, last64);
    const unsigned char* p = kf8 + ((long long)gi * 4 * 64 + lane) * 16;
#pragma unroll
    for (int a = 0; a < 4; ++a) kraw[g][a] = *reinterpret_cast<const u32x4_t*>(p + a * 1024);
    ks4[g] = *reinterpret_cast<const unsigned*>(ksc + gi * 64 + fr * 4);
  }
  __builtin_amdgcn_sched_barrier(0);
  // the query side behind the K requests (see attn_mfma_block)
  asm volatile("" ::"s"(q), "s"(fq.u), "s"(fq.stats), "s"(fq.n_part), "s"(fq.gw), "s"(fq.cb), "s"(fq.d));
  QRaw<T> qr;
  if constexpr (FQ) {
    fq_request<T>(qr, fq, fq_b, fq_h, lane);
  } else {
    qf[0] = *reinterpret_cast<const u32x4_t*>(q + kq * 16);
    qf[1] = *reinterpret_cast<const u32x4_t*>(q + kq * 16 + 8);
  }
#pragma unroll
  for (int g = 0; g < G; ++g) {  // 64 keys x 64 dims = 4 dim tiles of 1 KiB; lane `lane` also fetches the scale of ITS key
    const int gi = min(g * NW + wave, last64);
    const unsigned char* p = vtf8 + ((long long)gi * 4 * 64 + lane) * 16;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) vraw[g][dt] = *reinterpret_cast<const u32x4_t*>(p + dt * 1024);
    vsb[g] = vsc[gi * 64 + lane];
  }
  __builtin_amdgcn_sched_barrier(0);
  if constexpr (FQ) {
    T* qrow = qst + wave * 64;
    fq_finish<T>(qr, fq, lane, qrow);
    qf[0] = *reinterpret_cast<const u32x4_t*>(qrow + kq * 16);
    qf[1] = *reinterpret_cast<const u32x4_t*>(qrow + kq * 16 + 8);
  }

  float m = -1.0e30f;
#pragma unroll
  for (int g = 0; g < G; ++g) {
    const int g64 = g * NW + wave;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      u32x4_t k0, k1;
      kv8_widen(kraw[g][a], __builtin_bit_cast(float, ((ks4[g] >> (8 * a)) & 0xffu) << 23), k0, k1);
      f32x4_t acc = f32x4_t{0.f, 0.f, 0.f, 0.f};
      acc = sk_mfma<T>(k0, qf[0], acc);
      acc = sk_mfma<T>(k1, qf[1], acc);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int t = g64 * 64 + a * 16 + kq * 4 + r;
        const float sv = (t < n_keys) ? acc[r] : -1.0e30f;
        m = fmaxf(m, sv);
        if (fr == 0 && t < n_bound) sc[t] = sv;
      }
    }
  }
  m = tw_xor32_max(tw_xor16_max(m));
  if (lane == 0) red[wave] = m;
  __syncthreads();
  float M = red[0];
#pragma unroll
  for (int w = 1; w < NW; ++w) M = fmaxf(M, red[w]);

  f32x4_t o[4];
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) o[dt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  float ls = 0.f;
#pragma unroll
  for (int g = 0; g < G; ++g) {
    const int g64 = g * NW + wave;
    const int t0 = g64 * 64 + lane;  // one key per lane
    float pr = 0.f;
    if (t0 < n_bound) {
      pr = (t0 < n_keys) ? expf(sc[t0] - M) : 0.f;
      sc[t0] = pr;                                                  // plain probability: alignment rows, normaliser
      scv[t0] = pr * __builtin_bit_cast(float, vsb[g] << 23);       // probability x the key's V scale: the P.V operand
    }
    ls += pr;
    u32x4_t v0[4], v1[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) kv8_widen(vraw[g][dt], 1.0f, v0[dt], v1[dt]);
#pragma unroll
    for (int j = 0; j < 2; ++j) {   // operand j of lane (kq, .): keys kq*16 + j*8 .. +7 of this 64-key group
      float pv[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int t = g64 * 64 + kq * 16 + j * 8 + e;
        pv[e] = (t < n_bound) ? scv[t] : 0.f;   // same wavefront wrote these: LDS operations of a wavefront stay in order
      }
      const u32x4_t pf = pack16<T>(pv);
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) o[dt] = sk_mfma<T>(j == 0 ? v0[dt] : v1[dt], pf, o[dt]);
    }
  }
  ls = tw_wave_sum(ls);
  float* wl = red + NW;
  float* wo = red + 2 * NW;
  if (lane == 0) wl[wave] = ls;
  if (fr == 0) {
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
      for (int r = 0; r < 4; ++r) wo[wave * 64 + dt * 16 + kq * 4 + r] = o[dt][r];
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
    out_base[tw_xt_index<T>(out_j, out_k0 + tid)] = (T)(v * inv);
  }
  return inv;
}

// q [B, H*64] row-major; kc / vc per (stream, head): `rows` (multiple of 64) keys of K fragment-major / V^T fragment-major
// NW wavefronts of 64 keys each: 1 / 2 wavefronts while the sequence is shorter than 64 / 128 positions (the host picks the
// 64-key bucket per step): no idle wavefronts to launch and to meet at the barriers of the cross-wavefront reductions
template <typename T, bool SINGLE, int NW>
__global__ __launch_bounds__(NW * 64) void dec_self_attn_kernel(const T* __restrict__ q, const T* __restrict__ kc,
                                                                 const T* __restrict__ vc, int rows, int H, int key_bound,
                                                                 T* __restrict__ out, const DecState* __restrict__ stt, int rows_streams) {
  // argument order: what the K / V^T / q request addresses need comes first (kernarg preload, see skinny_mfma_kernel)
  __shared__ float sc[512];
  __shared__ float red[2 * 4 + 4 * 64];
  asm volatile("" ::"s"(q), "s"(kc), "s"(vc), "s"(rows), "s"(H), "s"(key_bound));
  const int h = blockIdx.x, b = blockIdx.y;
  // rows mode (prefill): row b is a position of stream b % rs - its K / V^T live in that stream's cache (the addresses depend on
  // kernel arguments only, as before) and it sees the keys up to ITS position (causal inside the launch)
  const int srow = rows_streams > 0 ? b % rows_streams : b;
  const int n_keys = stt->pos + (rows_streams > 0 ? b / rows_streams : 0) + 1;
  const long long base = ((long long)srow * H + h) * rows * 64;
  // the host guarantees pos < key_bound (a multiple of 64, <= rows): the requests do not wait for `pos`
  attn_mfma_block<T, NW, SINGLE>(q + ((long long)b * H + h) * 64, kc + base, vc + base, n_keys, key_bound, sc, red,
                                out + (long long)(b >> 4) * 16 * H * 64, b & 15, h * 64);  // fragment-major, groups of 16 streams
}

template <typename T, bool SINGLE, bool FQ>
__global__ __launch_bounds__(512) void dec_cross_attn_kernel(const T* __restrict__ ck, const T* __restrict__ cv, int H, int Tp,
                                                              int Tlen, int rows_streams, const T* __restrict__ q, T* __restrict__ out,
                                                              const int* __restrict__ align_slot, float* __restrict__ align,
                                                              int Ha, int P, const DecState* __restrict__ stt, FusedQ fq) {
  // (rows_streams sits among the LEADING arguments since round 5: the K / V^T request addresses depend on it - which stream's arena -
  //  and an argument behind the FusedQ block is not delivered with the wave: the requests waited for a scalar load)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* sc = reinterpret_cast<float*>(smem);  // [Tp] scores -> unnormalised probabilities
  __shared__ float red[2 * 8 + 8 * 64];
  __shared__ __attribute__((aligned(16))) T qst[FQ ? 8 * 64 : 8];
  asm volatile("" ::"s"(ck), "s"(cv), "s"(H), "s"(Tp), "s"(Tlen), "s"(rows_streams));
  const int h = blockIdx.x, b = blockIdx.y;
  const int srow = rows_streams > 0 ? b % rows_streams : b;   // rows mode (prefill): the stream whose encoder K / V this row attends to
  const long long base = ((long long)srow * H + h) * Tp * 64;
  const float inv = attn_mfma_block<T, 8, SINGLE, FQ>(q + ((long long)b * H + h) * 64, ck + base, cv + base, Tlen, Tp, sc, red,
                                                      out + (long long)(b >> 4) * 16 * H * 64, b & 15, h * 64, fq, b, h, qst);
  const int slot = align_slot ? align_slot[h] : -1;
  if (slot >= 0) {  // A11 side output: softmax row of an alignment head
    __syncthreads();
    const int prow = stt->pos + (rows_streams > 0 ? b / rows_streams : 0);
    float* row = align + (((long long)srow * Ha + slot) * P + prow) * Tlen;
    for (int t = threadIdx.x; t < Tlen; t += 512) row[t] = sc[t] * inv;
  }
}

template <int G, bool FQ>
__global__ __launch_bounds__(512) void dec_cross_attn_kv8_kernel(const unsigned char* __restrict__ ck, const unsigned char* __restrict__ cv,
                                                                  const unsigned char* __restrict__ ksc,
                                                                  const unsigned char* __restrict__ vsc, int H, int Tp, int Tlen,
                                                                  int rows_streams, const bf16_t* __restrict__ q, bf16_t* __restrict__ out,
                                                                  const int* __restrict__ align_slot, float* __restrict__ align,
                                                                  int Ha, int P, const DecState* __restrict__ stt, FusedQ fq) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* sc = reinterpret_cast<float*>(smem);  // [Tp] scores -> unnormalised probabilities
  float* scv = sc + Tp;                        // [Tp] probabilities x V scale
  __shared__ float red[2 * 8 + 8 * 64];
  __shared__ __attribute__((aligned(16))) bf16_t qst[FQ ? 8 * 64 : 8];
  asm volatile("" ::"s"(ck), "s"(cv), "s"(ksc), "s"(vsc), "s"(H), "s"(Tp), "s"(Tlen), "s"(rows_streams));
  const int h = blockIdx.x, b = blockIdx.y;
  const int srow = rows_streams > 0 ? b % rows_streams : b;   // rows mode (prefill), as in dec_cross_attn_kernel
  const long long hb = ((long long)srow * H + h) * Tp;
  const float inv = attn_mfma_block_kv8<8, G, FQ>(q + ((long long)b * H + h) * 64, ck + hb * 64, cv + hb * 64, ksc + hb, vsc + hb,
                                                  Tlen, Tp, sc, scv, red, out + (long long)(b >> 4) * 16 * H * 64, b & 15, h * 64,
                                                  fq, b, h, qst);
  const int slot = align_slot ? align_slot[h] : -1;
  if (slot >= 0) {  // A11 side output: softmax row of an alignment head
    __syncthreads();
    const int prow = stt->pos + (rows_streams > 0 ? b / rows_streams : 0);
    float* row = align + (((long long)srow * Ha + slot) * P + prow) * Tlen;
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

// static part of the mask (SuppressTokensLogitsProcessor list, HF:generation/logits_process.py:1869-1906) as a bitmap
__global__ void suppress_bitmap_kernel(const int* __restrict__ list, int n, unsigned* __restrict__ bits, int V) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    const int v = list[i];
    if (v >= 0 && v < V) atomicOr(&bits[v >> 5], 1u << (v & 31));
  }
}

// Sampling = Whisper's logits processors + first-index argmax (HF:generation/logits_process.py:1816-2047,
// HF:generation/utils.py:2925), in two launches:
//  * sampler_part_kernel: SAMPLER_NS workgroups per stream, each owning a slice of the vocabulary (one CU cannot evaluate
//    the mask and the comparisons for 51866 logits in less than ~20 us - the work is VALU-, not memory-bound).  The slice
//    is requested before anything else (addresses depend on kernel arguments only) and kept in registers for both
//    passes.  Output per slice: best text token, best timestamp token, sum of exp(x - slice max) over timestamp tokens.
//  * sampler_finish_kernel: ONE workgroup, wavefront b merges the slices of stream b (log-sum-exp merge), applies the
//    "timestamp mass beats every text token" rule, appends the token, and thread 0 advances the position afterwards.
// 32 slices of <= 4 x 512 logits per stream (one stream would otherwise put the whole vocabulary on 8 CUs: 10 us of a turbo step's
// 220; at 16 streams 512 short workgroups also beat 128 long ones, see launch_sampler)
constexpr int SAMPLER_NS_MAX = 32;

struct SamplerMask {  // dynamic part of the mask: uniform scalars derived from the decoding state
  int b_lo, b_hi, c_lo, c_hi, d_hi, e_lo, ts_begin;
  bool mask_eos, first, in_prompt, fin;
};

// b = row of the launch: a stream at the device position (ordinary step), or - rows mode, SamplerArgs::rows_streams > 0 - stream
// b % rs at the host-known position row_pos0 + b / rs, sampled with the state the loop would have had there (draft-and-verify)
__device__ __forceinline__ SamplerMask sampler_mask(const SamplerArgs& a, int b) {
  SamplerMask k;
  const int V = a.V;
  const bool rows = a.rows_streams > 0;
  const int stream = rows ? b % a.rows_streams : b;
  const int pos = rows ? a.row_pos0 + b / a.rows_streams : a.stt->pos;
  const int n_prompt = a.stt->n_prompt;
  const int cur_len = pos + 1;
  const int* seq = a.seq + (long long)stream * a.seq_ld;
  k.in_prompt = cur_len < n_prompt;     // still consuming the forced prompt
  k.fin = rows ? false : a.finished[b] != 0;           // HF: finished rows keep receiving pad_token_id
  k.first = (cur_len == n_prompt);
  const int n_new = cur_len - n_prompt;
  k.ts_begin = a.timestamps ? a.no_ts_id + 1 : V;
  bool last_ts = false, penult_ts = true;
  int lastts_tok = -1;
  if (a.timestamps && !k.in_prompt) {
    last_ts = (n_new >= 1) && (seq[cur_len - 1] >= k.ts_begin);
    penult_ts = (n_new < 2) || (seq[cur_len - 2] >= k.ts_begin);
    lastts_tok = rows ? a.row_lastts[b] : a.last_ts[b];
  }
  k.b_lo = 0; k.b_hi = 0;  // range masked by the pairing rule
  if (last_ts) {
    if (penult_ts) { k.b_lo = k.ts_begin; k.b_hi = V; } else { k.b_lo = 0; k.b_hi = a.eos; }
  }
  k.c_lo = 0; k.c_hi = 0;  // non-decreasing timestamps
  if (a.timestamps && lastts_tok >= 0) {
    k.c_lo = k.ts_begin;
    k.c_hi = (last_ts && !penult_ts) ? lastts_tok : lastts_tok + 1;
  }
  k.d_hi = 0; k.e_lo = V;  // first sampled token must be a timestamp <= max_initial
  if (a.timestamps && k.first) {
    k.d_hi = k.ts_begin;
    if (a.max_initial_ts >= 0) k.e_lo = k.ts_begin + a.max_initial_ts + 1;
  }
  k.mask_eos = n_new < a.min_new;
  return k;
}

template <int SAMPLER_NS, int SAMPLER_IT>
__global__ __launch_bounds__(256) void sampler_part_kernel(SamplerArgs a) {
  __shared__ MaxIdx red_text[4], red_ts[4];
  __shared__ float red_sum[4];
  const int part = blockIdx.x, b = blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int V = a.V;
  const float* lg = a.logits + (long long)b * V;
  const int chunk = ((V + 2 * SAMPLER_NS - 1) / (2 * SAMPLER_NS)) * 2;  // even slice length
  const int v0 = part * chunk, v1 = min(v0 + chunk, V);
  const bool vec_ok = (chunk <= SAMPLER_IT * 512) && ((V & 1) == 0) && V >= 2;
  const int v_last = max(V - 2, 0) & ~1;
  float s0[SAMPLER_IT], s1[SAMPLER_IT];
  unsigned wb[SAMPLER_IT];
#pragma unroll
  for (int it = 0; it < SAMPLER_IT; ++it) {  // unconditional (clamped) requests; the pair (v, v+1) shares one bitmap word
    const int v = min(v0 + (it * 256 + tid) * 2, v_last);
    const float2 x2 = *reinterpret_cast<const float2*>(lg + v);
    s0[it] = x2.x;
    s1[it] = x2.y;
    wb[it] = a.suppress_bits[v >> 5];
  }
  __builtin_amdgcn_sched_barrier(0);
  const SamplerMask k = sampler_mask(a, b);
  auto masked_at = [&](int v, unsigned word) -> bool {
    bool masked = ((word >> (v & 31)) & 1u) != 0;
    if (k.mask_eos && v == a.eos) masked = true;
    if (a.timestamps && v == a.no_ts_id) masked = true;
    if (v >= k.b_lo && v < k.b_hi) masked = true;
    if (v >= k.c_lo && v < k.c_hi) masked = true;
    if (v < k.d_hi) masked = true;
    if (v >= k.e_lo) masked = true;
    if (k.first)
      for (int i = 0; i < a.n_begin_suppress; ++i) masked |= (v == a.begin_suppress[i]);
    return masked;
  };
  // ---- pass 1: best text token, best timestamp token of the slice ----
  MaxIdx bt{-INFINITY, 0x7fffffff}, bs{-INFINITY, 0x7fffffff};
  if (vec_ok) {
#pragma unroll
    for (int it = 0; it < SAMPLER_IT; ++it) {
      const int v = v0 + (it * 256 + tid) * 2;
      const bool in = v < v1;
      const int vc = in ? v : 0;
      s0[it] = (in && !masked_at(vc, wb[it])) ? s0[it] : -INFINITY;
      s1[it] = (in && !masked_at(vc + 1, wb[it])) ? s1[it] : -INFINITY;
      if (in) {
        MaxIdx c0{s0[it], v}, c1{s1[it], v + 1};
        if (v < k.ts_begin) bt = better(bt, c0); else bs = better(bs, c0);
        if (v + 1 < k.ts_begin) bt = better(bt, c1); else bs = better(bs, c1);
      }
    }
  } else {
    for (int v = v0 + tid; v < v1; v += 256) {
      MaxIdx c{masked_at(v, a.suppress_bits[v >> 5]) ? -INFINITY : lg[v], v};
      if (v < k.ts_begin) bt = better(bt, c); else bs = better(bs, c);
    }
  }
  bt = wave_best(bt);
  bs = wave_best(bs);
  if (lane == 0) { red_text[wave] = bt; red_ts[wave] = bs; }
  __syncthreads();
  bt = red_text[0];
  bs = red_ts[0];
  for (int w = 1; w < 4; ++w) { bt = better(bt, red_text[w]); bs = better(bs, red_ts[w]); }
  // ---- pass 2: sum of exp(x - slice max) over the slice's timestamp tokens ----
  float sum = 0.f;
  if (a.timestamps && bs.v > -INFINITY) {
    if (vec_ok) {
#pragma unroll
      for (int it = 0; it < SAMPLER_IT; ++it) {
        const int v = v0 + (it * 256 + tid) * 2;
        if (v < v1) {
          if (v >= k.ts_begin) sum += expf(s0[it] - bs.v);
          if (v + 1 >= k.ts_begin) sum += expf(s1[it] - bs.v);
        }
      }
    } else {
      for (int v = max(v0, k.ts_begin) + tid; v < v1; v += 256)
        sum += masked_at(v, a.suppress_bits[v >> 5]) ? 0.f : expf(lg[v] - bs.v);
    }
  }
  sum = wave_sum(sum);
  if (lane == 0) red_sum[wave] = sum;
  __syncthreads();
  if (tid == 0) {
    SamplerPartial o;
    o.bt_v = bt.v; o.bt_i = bt.i; o.bs_v = bs.v; o.bs_i = bs.i;
    o.sum = red_sum[0] + red_sum[1] + red_sum[2] + red_sum[3];
    a.partials[b * SAMPLER_NS_MAX + part] = o;
  }
}

// wavefront b <- stream b; thread 0 advances the position once every wavefront has used it
// the next step's input row of stream b (A6: HF:models/whisper/modeling_whisper.py:733-771), by one wavefront: 16-byte vectors of the
// token row and the positional row, added in float32, rounded once, stored where tw_xt_index puts elements 8v .. 8v+7 (E per vector)
template <typename T>
__device__ __forceinline__ void embed_row(const SamplerArgs& a, int b, int id, int p, int lane) {
  constexpr int E = ElemTraits<T>::kPer16B;
  const T* tok = reinterpret_cast<const T*>(a.tok_emb) + (long long)id * a.d;
  const T* pos = reinterpret_cast<const T*>(a.pos_emb) + (long long)p * a.d;
  T* x = reinterpret_cast<T*>(a.x_next) + (long long)(b >> 4) * 16 * a.d;
  for (int v = lane; v * E < a.d; v += 64) {
    const u32x4_t tv = *reinterpret_cast<const u32x4_t*>(tok + v * E), pv = *reinterpret_cast<const u32x4_t*>(pos + v * E);
    T o[E];
    const T* tp = reinterpret_cast<const T*>(&tv);
    const T* pp = reinterpret_cast<const T*>(&pv);
#pragma unroll
    for (int e = 0; e < E; ++e) o[e] = (T)((float)tp[e] + (float)pp[e]);
    *reinterpret_cast<u32x4_t*>(x + tw_xt_index<T>(b & 15, v * E)) = *reinterpret_cast<const u32x4_t*>(o);
  }
}

// merge of the vocabulary slices of row b by one wavefront (all 64 lanes): the token HF's processors + argmax pick, in every lane
__device__ __forceinline__ int sampler_merge(const SamplerArgs& a, int b, int lane) {
  SamplerPartial p = a.partials[b * SAMPLER_NS_MAX + min(lane, a.n_slices - 1)];
  const bool on = lane < a.n_slices;
  MaxIdx bt{on ? p.bt_v : -INFINITY, on ? p.bt_i : 0x7fffffff}, bs{on ? p.bs_v : -INFINITY, on ? p.bs_i : 0x7fffffff};
  const float my_m = bs.v;
  bt = wave_best(bt);
  bs = wave_best(bs);
  float part_sum = (on && my_m > -INFINITY) ? p.sum * expf(my_m - bs.v) : 0.f;  // log-sum-exp merge of the slices
  const float tot = wave_sum(part_sum);
  bool force_ts = false;
  if (a.timestamps && bs.v > -INFINITY) force_ts = (bs.v + logf(tot)) > bt.v;
  int choice;
  if (force_ts) choice = bs.i;
  else choice = (bs.v > bt.v) ? bs.i : bt.i;
  if (choice == 0x7fffffff) choice = 0;  // everything masked: torch.argmax of all -inf is 0
  return choice;
}

__global__ __launch_bounds__(1024) void sampler_finish_kernel(SamplerArgs a) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int pos_now = a.stt->pos;              // (thread 0 advances it behind the barrier at the end)
  for (int b = tid >> 6; b < a.B; b += 16) {  // wavefront w <- streams w, w+16, ...
    int next_id = 0;
    const SamplerMask k = sampler_mask(a, b);
    const int choice = sampler_merge(a, b, lane);
    if (lane == 0) {
      const int cur_len = a.stt->pos + 1;
      int* seq = a.seq + (long long)b * a.seq_ld;
      if (k.in_prompt) {
        next_id = seq[cur_len];
        a.cur_ids[b] = next_id;
      } else if (k.fin) {
        seq[cur_len] = a.pad;
        a.cur_ids[b] = a.pad;
        next_id = a.pad;
      } else {
        seq[cur_len] = choice;
        a.cur_ids[b] = choice;
        next_id = choice;
        if (a.timestamps && choice >= k.ts_begin) a.last_ts[b] = choice;
        if (choice == a.eos) a.finished[b] = 1;
      }
    }
    if (a.x_next) {   // (kernel-uniform) the token just appended is the next step's input at position pos + 1
      const int id = __builtin_amdgcn_readfirstlane(next_id);   // lane 0's
      if (a.dtype == 1) embed_row<bf16_t>(a, b, id, pos_now + 1, lane);
      else if (a.dtype == 2) embed_row<f16_t>(a, b, id, pos_now + 1, lane);
      else embed_row<float>(a, b, id, pos_now + 1, lane);
    }
  }
  __syncthreads();
  if (tid == 0) a.stt->pos += 1;
}

// Rows mode (draft-and-verify; SamplerArgs): every row's token, the first position at which a stream's token differs from the token the
// NEXT row was given, and - once the round is decided (a mismatch, or the round's last launch) - the state the ordinary loop would have
// after producing the token at p_acc + 1 itself: that token appended to every stream's history (seq, cur_ids, last_ts, finished), the
// device position = p_acc + 1.  Everything at positions > p_acc + 1 (self-attention K / V rows, alignment rows, stale given tokens) is
// overwritten by the steps that follow before anything reads it.  One workgroup; rows = streams x positions <= 64.
__global__ __launch_bounds__(1024) void sampler_rows_finish_kernel(SamplerArgs a) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int rs = a.rows_streams;
  for (int r = tid >> 6; r < a.B; r += 16) {
    const int stream = r % rs, p = a.row_pos0 + r / rs;
    const SamplerMask k = sampler_mask(a, r);
    int choice = sampler_merge(a, r, lane);
    if (lane == 0) {
      const int* seq = a.seq + (long long)stream * a.seq_ld;
      if (k.in_prompt) choice = seq[p + 1];          // the prompt is not sampled
      a.row_choice[r] = choice;
      if (p < a.row_last_pos && choice != seq[p + 1]) atomicMin(&a.verify_state[0], p);
    }
  }
  __syncthreads();
  __shared__ int p_acc_s;
  if (tid == 0) {
    const int fm = a.verify_state[0];     // (this workgroup's own atomics, behind the barrier)
    int p_acc = -1;
    if (fm != 0x7fffffff) p_acc = fm;
    else if (a.row_is_last) p_acc = a.row_last_pos;
    p_acc_s = p_acc;
  }
  __syncthreads();
  const int p_acc = p_acc_s;
  if (p_acc < 0) {                        // undecided: the next launch of the round continues behind this launch's positions
    if (tid == 0) a.stt->pos += a.B / rs;
    return;
  }
  const int n_streams = rs;
  if (tid < n_streams) {
    const int b = tid;
    const int n_prompt = a.stt->n_prompt;
    const int ts_begin = a.timestamps ? a.no_ts_id + 1 : a.V;
    int* seq = a.seq + (long long)b * a.seq_ld;
    const int tok = a.row_choice[(p_acc - a.row_pos0) * rs + b];   // (a mismatch can only be found in THIS launch: earlier ones ended the round)
    int lt = -1;
    for (int i = n_prompt; i <= p_acc; ++i) if (a.timestamps && seq[i] >= ts_begin) lt = seq[i];
    if (p_acc + 1 >= n_prompt) {          // a sampled token (not a prompt token)
      if (a.timestamps && tok >= ts_begin) lt = tok;
      a.finished[b] = (tok == a.eos) ? 1 : 0;
    } else {
      a.finished[b] = 0;
    }
    seq[p_acc + 1] = tok;
    a.cur_ids[b] = tok;
    a.last_ts[b] = lt;
    a.verify_state[2 + b] = tok;
  }
  if (tid == 0) {
    a.verify_state[1] = p_acc + 1;
    a.stt->pos = p_acc + 1;
  }
}

__global__ void advance_kernel(DecState* stt, int n) { stt->pos += n; }

}  // namespace

hipError_t init_decode_kernels() { return hipSuccess; }  // nothing to configure (kept for the call site in tw_create)

static int env_int(const char* name, int dflt) {
  const char* e = getenv(name);
  return e ? atoi(e) : dflt;
}

// how launches for more than 16 streams move their operands (skinny_mfma_kernel's MODE): 2 = operand rings where the shape allows
// (default), 0 = plain rounds (rounds 3-4); TW_SK_CG_MODE for A/B runs
static int cg_mode() {
  static const int m = env_int("TW_SK_CG_MODE", 3);   // 3 = 2 + two tiles per workgroup (DUAL) for the launches with more tiles than TW_SK_RING_BLOCKS
  return m;
}

template <typename T, int NW, int SK_MAXS, bool MULTI, bool W8, int CG, int TR, bool A16 = false, int MODE = 0, bool SPLIT2 = false>
static hipError_t skinny_launch_cg(const GemvArgs& a, dim3 grid, size_t lds1, hipStream_t st) {
  const bool ln = a.ln_gw != nullptr;
  const size_t lds = lds1 * CG;
#define SK_GO(LNV, EPIV) hipLaunchKernelGGL((skinny_mfma_kernel<T, NW, SK_MAXS, LNV, EPIV, MULTI, W8, CG, TR, A16, MODE, SPLIT2>), grid, dim3(NW * 64), lds, st, \
                                            a.x, a.W, a.K, a.B, a.N, a.rg, a.wscale, a.bias, a.res, a.ln_gw, a)
  if constexpr (SPLIT2) {  // the 8-wavefront flavour of the long-K residual projection (fc2 above 16 rows): kernel header
    if (ln || a.y_f32 || a.kcache || a.gelu) return hipErrorInvalidValue;
    if constexpr (MULTI || NW != 8 || CG == 1) {
      return hipErrorInvalidValue;
    } else {
      if (a.res) SK_GO(false, SK_RES);
      else SK_GO(false, SK_STORE);
    }
  } else if constexpr (TR != 16) {  // narrow tiles: plain / residual / GELU projections, one tile per workgroup
    if (a.y_f32 || a.kcache) return hipErrorInvalidValue;
    if constexpr (MULTI || (W8 && TR != 8)) {
      return hipErrorInvalidValue;
    } else if constexpr (NW >= 16) {
      if (ln || a.gelu) return hipErrorInvalidValue;
      if (a.res) SK_GO(false, SK_RES);
      else SK_GO(false, SK_STORE);
    } else {
      if (a.gelu) { if (!ln || a.res) return hipErrorInvalidValue; SK_GO(true, SK_GELU); }
      else if (a.res) { if (ln) return hipErrorInvalidValue; SK_GO(false, SK_RES); }
      else if (ln) SK_GO(true, SK_STORE);
      else SK_GO(false, SK_STORE);
    }
  } else if constexpr (NW >= 16) {  // 16 wavefronts per tile only for the long-K residual projections (fc2)
    if (ln || a.y_f32 || a.kcache || a.gelu) return hipErrorInvalidValue;
    if (a.res) SK_GO(false, SK_RES);
    else SK_GO(false, SK_STORE);
  } else if (a.y_f32) {
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

template <typename T, int NW, int SK_MAXS, bool MULTI, bool W8, int TR, bool SPLIT2 = false>
static hipError_t skinny_launch_v(const GemvArgs& a, dim3 grid, size_t lds1, hipStream_t st) {
  if constexpr (W8) {
    // MXFP8 weights.  a.a16: W8A16 (weights widened in registers, bf16 activations) - the only flavour with several groups
    // of 16 streams (2 steps = 8 activation fragments per group in flight, further rounds for longer K)
    if (a.a16) {
      if (a.B <= 16) {
        if constexpr (SPLIT2) return hipErrorInvalidValue;
        else return skinny_launch_cg<T, NW, SK_MAXS, MULTI, true, 1, TR, true>(a, grid, lds1, st);
      }
      if constexpr (NW != 8 || SK_MAXS != 2) {
        return hipErrorInvalidValue;
      } else {
        if (a.B <= 32) return skinny_launch_cg<T, NW, SK_MAXS, MULTI, true, 2, TR, true, 0, SPLIT2>(a, grid, lds1, st);
        return skinny_launch_cg<T, NW, SK_MAXS, MULTI, true, 4, TR, true, 0, SPLIT2>(a, grid, lds1, st);
      }
    }
    if (a.B > 16 || SPLIT2) return hipErrorInvalidValue;
    if constexpr (!SPLIT2) return skinny_launch_cg<T, NW, SK_MAXS, MULTI, true, 1, TR>(a, grid, lds1, st);
    return hipErrorInvalidValue;
  } else {
    if (a.B <= 16) {
      if constexpr (SPLIT2) return hipErrorInvalidValue;
      else return skinny_launch_cg<T, NW, SK_MAXS, MULTI, W8, 1, TR>(a, grid, lds1, st);
    }
    // several groups of 16 streams: 8 wavefronts x 5 fragments in flight (more rounds for long K) keeps the per-group
    // activation fragments inside the register file
    if constexpr (NW != 8 || SK_MAXS != 5) {
      return hipErrorInvalidValue;
    } else {
      if (a.B <= 32) return skinny_launch_cg<T, NW, SK_MAXS, MULTI, W8, 2, TR, false, 0, SPLIT2>(a, grid, lds1, st);
      return skinny_launch_cg<T, NW, SK_MAXS, MULTI, W8, 4, TR, false, 0, SPLIT2>(a, grid, lds1, st);
    }
  }
}

// split2: this launch is the several-groups form of a projection whose one-group form runs on 16 wavefronts (skinny_launch: `longk`)
template <typename T, int NW, int TR>
static hipError_t skinny_launch_nw(const GemvArgs& a0, hipStream_t st, bool split2 = false) {
  constexpr int E = ElemTraits<T>::kPer16B;
  GemvArgs a = a0;
  if (a.K % (4 * E) != 0 || a.B > 64) return hipErrorInvalidValue;
  const size_t lds = (size_t)NW * kRedTile * 4;
  const int tiles = (a.N + TR - 1) / TR;
  // at most `max_blocks` workgroups: tall matrices (the tied logits projection) walk several tiles per workgroup
  static const int max_blocks = env_int("TW_SK_MAX_BLOCKS", 512);
  const int steps_per_wave = (a.K / E / 4 + NW - 1) / NW;
  const bool groups = a.B > 16;  // several groups of 16 streams: 5 fragments in flight, further rounds for longer K
  a.rg = (tiles + max_blocks - 1) / max_blocks;
  if (a.rg < 1 || steps_per_wave > (groups ? 5 : 10)) a.rg = 1;  // several tiles per workgroup only with one round per tile
  dim3 grid((tiles + a.rg - 1) / a.rg);
  if constexpr (NW == 8) {
    // more than 16 streams, K an exact multiple of the wavefronts' step: operand rings (skinny_mfma_kernel MODE 2), one tile per
    // workgroup - or, for the launches with more 16-row tiles than the decode loop has compute units (QKV, fc1: 320 on 160 CUs),
    // TWO tiles per workgroup contracted against one sweep of the activation rings (DUAL: half the activation traffic)
    const int steps = a.K / E / 4;
    static const int ring_blocks = env_int("TW_SK_RING_BLOCKS", 160);
    if (groups && a.rg == 1 && cg_mode() >= 2 && steps % NW == 0 && !a.y_f32) {
      const int spw = steps / NW;
      bool dual = false;
      if constexpr (TR == 16)
        dual = cg_mode() >= 3 && !split2 && (a.kcache || a.gelu) && ring_blocks > 0 && tiles > ring_blocks && tiles <= 2 * ring_blocks && tiles % 2 == 0 && spw == (E == 8 ? 5 : 10);   // (exactly the shapes instantiated below)
      if (dual) { a.rg = 2; grid = dim3(tiles / 2); }
#define SK_RING(SPW, SP2, MU) (a.B <= 32 ? skinny_launch_cg<T, NW, SPW, MU, false, 2, TR, false, 2, SP2>(a, grid, lds, st) \
                                         : skinny_launch_cg<T, NW, SPW, MU, false, 4, TR, false, 2, SP2>(a, grid, lds, st))
      if constexpr (E == 8) {          // 16-bit contexts: K = 1280 / 5120
        if constexpr (TR == 16) { if (dual && spw == 5) return SK_RING(5, false, true); }
        if (spw == 5 && !split2) return SK_RING(5, false, false);
        if (spw == 20) return split2 ? SK_RING(20, true, false) : SK_RING(20, false, false);
      } else {                         // strict-f32 contexts
        if constexpr (TR == 16) { if (dual && spw == 10) return SK_RING(10, false, true); }
        if (spw == 10 && !split2) return SK_RING(10, false, false);
        if (spw == 40) return split2 ? SK_RING(40, true, false) : SK_RING(40, false, false);
      }
#undef SK_RING
    }
    if (split2) {   // other long K (e.g. ffn = 4096): rounds of 5 fragments, the slice's midpoint found per step
      if (!groups || a.rg != 1) return hipErrorInvalidValue;
      return skinny_launch_v<T, NW, 5, false, false, TR, true>(a, grid, lds, st);
    }
  } else if (split2) {
    return hipErrorInvalidValue;
  }
  if constexpr (TR != 16) {
    if (a.rg > 1) return hipErrorInvalidValue;
    if (steps_per_wave <= 5 || groups) return skinny_launch_v<T, NW, 5, false, false, TR>(a, grid, lds, st);
    return skinny_launch_v<T, NW, 10, false, false, TR>(a, grid, lds, st);
  } else {
    if (a.rg > 1) {
      if (steps_per_wave <= 5 || groups) return skinny_launch_v<T, NW, 5, true, false, 16>(a, grid, lds, st);
      return skinny_launch_v<T, NW, 10, true, false, 16>(a, grid, lds, st);
    }
    if (steps_per_wave <= 5 || groups) return skinny_launch_v<T, NW, 5, false, false, 16>(a, grid, lds, st);
    return skinny_launch_v<T, NW, 10, false, false, 16>(a, grid, lds, st);
  }
}

// MXFP8 weights: 128-k steps; 2 steps per wavefront cover K = 1280 with 8 wavefronts, 3 cover K = 5120 with 16
template <int NW, int TR>
static hipError_t skinny_launch_w8(const GemvArgs& a0, hipStream_t st, bool split2 = false) {
  GemvArgs a = a0;
  if (a.K % 128 != 0 || a.B > 64) return hipErrorInvalidValue;
  const size_t lds = (size_t)NW * kRedTile * 4;
  const int tiles = (a.N + TR - 1) / TR;
  static const int max_blocks = env_int("TW_SK_MAX_BLOCKS", 512);
  const int steps_per_wave = (a.K / 128 + NW - 1) / NW;
  const bool groups = a.B > 16;   // W8A16 only (skinny_launch_v): 2 steps in flight per group, further rounds for longer K
  // several groups of streams: every workgroup re-reads the whole activation block (164 KB per 64 streams at K = 1280, more bytes than
  // its weight tile), so the 320-tile launches walk TWO tiles per workgroup and keep the activation fragments in registers
  // (64 streams x 15 s: 3.50 -> 3.32 ms per step, profiles/r04_b64_15s_two_tiles_per_workgroup.txt; neutral for the bf16 kernels)
  static const int max_blocks_groups = env_int("TW_SK_MAX_BLOCKS_W8_GROUPS", 160);
  a.rg = (tiles + (groups ? max_blocks_groups : max_blocks) - 1) / (groups ? max_blocks_groups : max_blocks);
  if (a.rg < 1 || steps_per_wave > (groups ? 2 : 3)) a.rg = 1;   // several tiles per workgroup only with one round per tile
  static const int dbg_mask = env_int("TW_DBG_W8_MULTI_MASK", 7);   // diagnostics: bit 0 QKV, 1 fc1, 2 logits may walk several tiles per workgroup
  if (groups && a.rg > 1 && !((a.kcache ? 1 : (a.gelu ? 2 : (a.y_f32 ? 4 : 0))) & dbg_mask)) a.rg = 1;
  dim3 grid((tiles + a.rg - 1) / a.rg);
  if (split2) {   // W8A16 above 16 rows, long K: rounds of 2 steps, two accumulator sets (skinny_mfma_kernel's header)
    if constexpr (NW != 8) {
      return hipErrorInvalidValue;
    } else {
      if (!groups || !a.a16 || a.rg != 1) return hipErrorInvalidValue;
      return skinny_launch_v<bf16_t, NW, 2, false, true, TR, true>(a, grid, lds, st);
    }
  }
  if constexpr (TR != 16) {
    if (a.rg > 1) return hipErrorInvalidValue;
    if (steps_per_wave <= 2 || groups) return skinny_launch_v<bf16_t, NW, 2, false, true, TR>(a, grid, lds, st);
    return skinny_launch_v<bf16_t, NW, 3, false, true, TR>(a, grid, lds, st);
  } else {
    if (a.rg > 1) {
      if (steps_per_wave <= 2 || groups) return skinny_launch_v<bf16_t, NW, 2, true, true, 16>(a, grid, lds, st);
      return skinny_launch_v<bf16_t, NW, 3, true, true, 16>(a, grid, lds, st);
    }
    if (steps_per_wave <= 2 || groups) return skinny_launch_v<bf16_t, NW, 2, false, true, 16>(a, grid, lds, st);
    return skinny_launch_v<bf16_t, NW, 3, false, true, 16>(a, grid, lds, st);
  }
}

template <typename T>
static hipError_t skinny_launch(const GemvArgs& a, hipStream_t st) {
  static const int nw_big = env_int("TW_SK_NW_BIGK", 16);  // wavefronts per tile when K is long (fc2: K = 5120)
  // long-K residual projection: 16 wavefronts per tile with one group of streams, 8 (two accumulator sets: the same sixteen K slices,
  // the same order of additions - skinny_mfma_kernel's header) with several
  const bool longk = a.K >= 4096 && nw_big == 16 && !a.ln_gw && !a.y_f32 && !a.kcache && !a.gelu;
  const bool big = longk && a.B <= 16, split2 = longk && a.B > 16;
  if (a.wscale) {
    if (ElemTraits<T>::kCode != 1) return hipErrorInvalidValue;
    if (a.tr == 8) return big ? skinny_launch_w8<16, 8>(a, st) : skinny_launch_w8<8, 8>(a, st, split2);
    if (a.tr != 0 && a.tr != 16) return hipErrorInvalidValue;
    return big ? skinny_launch_w8<16, 16>(a, st) : skinny_launch_w8<8, 16>(a, st, split2);
  }
  static const int nw_narrow = env_int("TW_SK_NW_NARROW", 8);  // wavefronts per 8-row tile of the N = 1280, K = 1280 projections (4: A/B)
  if (a.tr == 8 && !longk && nw_narrow == 4 && a.B <= 16) return skinny_launch_nw<T, 4, 8>(a, st);
  if (a.tr == 8) return big ? skinny_launch_nw<T, 16, 8>(a, st) : skinny_launch_nw<T, 8, 8>(a, st, split2);
  if (a.tr == 4) return big ? skinny_launch_nw<T, 16, 4>(a, st) : skinny_launch_nw<T, 8, 4>(a, st, split2);
  if (a.tr != 0 && a.tr != 16) return hipErrorInvalidValue;
  return big ? skinny_launch_nw<T, 16, 16>(a, st) : skinny_launch_nw<T, 8, 16>(a, st, split2);
}

template <typename T>
static hipError_t gemv_b(const GemvArgs& a0, hipStream_t st) {
  if (a0.B < 1 || a0.B > 64) return hipErrorInvalidValue;
  GemvArgs a = a0;
  if (a.kcache) {   // QKV launch: a fourth segment of d_model rows only together with its destination
    if ((a.N == 4 * a.d_model) != (a.u != nullptr)) return hipErrorInvalidValue;
    a.nsplit = a.N;
  } else if (a.u) {  // residual launch with a composed half
    if (!a.res || !a.stats || a.nsplit <= 0 || a.nsplit >= a.N || a.nsplit % 16 != 0) return hipErrorInvalidValue;
  } else {
    a.nsplit = a.N;
  }
  return skinny_launch<T>(a, st);
}

hipError_t launch_gemv(int dtype, const GemvArgs& a, hipStream_t st) {
  return dtype == 1 ? gemv_b<bf16_t>(a, st) : (dtype == 2 ? gemv_b<f16_t>(a, st) : gemv_b<float>(a, st));
}

hipError_t launch_dec_self_attn(int dtype, const void* q, const void* kc, const void* vc, int rows, void* out, int B, int H,
                                int key_bound, const DecState* stt, int rows_streams, hipStream_t st) {
  // key_bound: an upper bound of pos+1 for this call known on the host (prompt + max new tokens)
  int kb = (key_bound + 63) / 64 * 64;
  if (rows % 64 != 0 || rows > 512 || kb > rows) return hipErrorInvalidValue;
  const bool single = kb <= 256;
  const int nw = kb <= 64 ? 1 : (kb <= 128 ? 2 : 4);
#define SA_GO(TT, SV, NWV) hipLaunchKernelGGL((dec_self_attn_kernel<TT, SV, NWV>), dim3(H, B), dim3(NWV * 64), 0, st, (const TT*)q, \
                                              (const TT*)kc, (const TT*)vc, rows, H, kb, (TT*)out, stt, rows_streams)
#define SA_PICK(TT) do { if (nw == 1) SA_GO(TT, true, 1); else if (nw == 2) SA_GO(TT, true, 2); else if (single) SA_GO(TT, true, 4); \
                         else SA_GO(TT, false, 4); } while (0)
  if (dtype == 1) SA_PICK(bf16_t); else if (dtype == 2) SA_PICK(f16_t); else SA_PICK(float);
#undef SA_PICK
#undef SA_GO
  return hipGetLastError();
}

hipError_t launch_dec_cross_attn(int dtype, const void* q, const FusedQ& fq, const void* ck, const void* cv, void* out, int B, int H,
                                 int T, int Tp, const int* align_slot_for_head, float* align, int Ha, int P,
                                 const DecState* stt, const unsigned char* ksc, const unsigned char* vsc, int rows_streams,
                                 hipStream_t st) {
  if (Tp % 64 != 0 || Tp < T) return hipErrorInvalidValue;
  const bool single = Tp <= 512;
  const bool f = fq.u != nullptr;
  if (f) {   // the partial statistics must fit the per-lane request budget of fq_request
    if (!fq.stats || !fq.gw || !fq.cb || fq.d != H * 64 || fq.n_part < 1 || fq.n_part > 192) return hipErrorInvalidValue;
  } else if (!q) {
    return hipErrorInvalidValue;
  }
  if (ksc || vsc) {  // fp8 K / V^T caches with per-key scales (TW_BF16_MXFP8 contexts)
    if (!ksc || !vsc || dtype != 1) return hipErrorInvalidValue;
    const size_t lds8 = (size_t)Tp * sizeof(float) * 2;
#define CA8_GO(GV, FV) hipLaunchKernelGGL((dec_cross_attn_kv8_kernel<GV, FV>), dim3(H, B), dim3(512), lds8, st,                        \
                                          (const unsigned char*)ck, (const unsigned char*)cv, ksc, vsc, H, Tp, T, rows_streams, (const bf16_t*)q,  \
                                          (bf16_t*)out, align_slot_for_head, align, Ha, P, stt, fq)
#define CA8_PICK(FV) do { if (Tp <= 512) CA8_GO(1, FV); else if (Tp <= 1024) CA8_GO(2, FV); else if (Tp <= 1536) CA8_GO(3, FV);      \
                          else return hipErrorInvalidValue; } while (0)
    if (f) CA8_PICK(true); else CA8_PICK(false);
#undef CA8_PICK
#undef CA8_GO
    return hipGetLastError();
  }
  const size_t lds = (size_t)Tp * sizeof(float);
#define CA_GO(TT, SV, FV) hipLaunchKernelGGL((dec_cross_attn_kernel<TT, SV, FV>), dim3(H, B), dim3(512), lds, st, (const TT*)ck,      \
                                             (const TT*)cv, H, Tp, T, rows_streams, (const TT*)q, (TT*)out, align_slot_for_head, align, Ha, P, stt, fq)
#define CA_PICK(TT) do { if (single) { if (f) CA_GO(TT, true, true); else CA_GO(TT, true, false); }                                  \
                         else { if (f) CA_GO(TT, false, true); else CA_GO(TT, false, false); } } while (0)
  if (dtype == 1) CA_PICK(bf16_t); else if (dtype == 2) CA_PICK(f16_t); else CA_PICK(float);
#undef CA_PICK
#undef CA_GO
  return hipGetLastError();
}

hipError_t launch_sampler(const SamplerArgs& a0, hipStream_t st) {
  if (a0.B < 1 || a0.B > 64 || !a0.partials || !a0.suppress_bits) return hipErrorInvalidValue;
  SamplerArgs a = a0;
  // 32 vocabulary slices per stream (<= 4 x 512 logits per workgroup).  Until round 3 launches for >= 8 streams used 8 slices of 13 x 512
  // ("8 x B workgroups cover the chip"); same-box A/B at 16 streams: 1.3824 -> 1.3759 ms per step with 32 (TW_SAMPLER_SLICES_MANY=8 restores it)
  static const int many = env_int("TW_SAMPLER_SLICES_MANY", 32);
  a.n_slices = a.B >= 8 ? (many == 8 ? 8 : 32) : 32;
  if (a.n_slices == 8) hipLaunchKernelGGL((sampler_part_kernel<8, 13>), dim3(8, a.B), dim3(256), 0, st, a);
  else hipLaunchKernelGGL((sampler_part_kernel<32, 4>), dim3(32, a.B), dim3(256), 0, st, a);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(sampler_finish_kernel, dim3(1), dim3(1024), 0, st, a);  // also advances the position
  return hipGetLastError();
}

hipError_t launch_sampler_rows(const SamplerArgs& a0, hipStream_t st) {
  if (a0.B < 1 || a0.B > 64 || !a0.partials || !a0.suppress_bits || a0.rows_streams < 1 || a0.B % a0.rows_streams != 0 || !a0.row_lastts ||
      !a0.row_choice || !a0.verify_state)
    return hipErrorInvalidValue;
  SamplerArgs a = a0;
  a.n_slices = 32;    // the slicing of launch_sampler (the partial sums are merged in the same order: same bits)
  a.x_next = nullptr;
  static const int many = env_int("TW_SAMPLER_SLICES_MANY", 32);
  if (many == 8) return hipErrorInvalidValue;   // (A/B switch of the step sampler: the two slicings sum in different orders)
  hipLaunchKernelGGL((sampler_part_kernel<32, 4>), dim3(32, a.B), dim3(256), 0, st, a);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(sampler_rows_finish_kernel, dim3(1), dim3(1024), 0, st, a);
  return hipGetLastError();
}

hipError_t launch_suppress_bitmap(const int* list, int n, unsigned* bits, int V, hipStream_t st) {
  hipError_t e = hipMemsetAsync(bits, 0, (size_t)((V + 31) / 32) * 4, st);
  if (e != hipSuccess || n <= 0) return e;
  hipLaunchKernelGGL(suppress_bitmap_kernel, dim3((n + 255) / 256), dim3(256), 0, st, list, n, bits, V);
  return hipGetLastError();
}

hipError_t launch_advance(DecState* stt, int n, hipStream_t st) {
  hipLaunchKernelGGL(advance_kernel, dim3(1), dim3(1), 0, st, stt, n);
  return hipGetLastError();
}

hipError_t launch_tile_weights(int dtype, const void* src, void* dst, int N, int K, int tr, hipStream_t st) {
  const int E = dtype == 0 ? 4 : 8;
  if (K % (4 * E) != 0 || (tr != 16 && tr != 8 && tr != 4) || N % tr != 0 && tr != 16) return hipErrorInvalidValue;
  const long long total = (long long)((N + tr - 1) / tr) * (K / E / 4) * 4 * tr;
  dim3 grid((unsigned)((total + 255) / 256));
  if (dtype != 0) hipLaunchKernelGGL(tile_weights_kernel<bf16_t>, grid, dim3(256), 0, st, (const bf16_t*)src, (bf16_t*)dst, N, K, tr);   // a pure copy: bf16 and f16 alike
  else hipLaunchKernelGGL(tile_weights_kernel<float>, grid, dim3(256), 0, st, (const float*)src, (float*)dst, N, K, tr);
  return hipGetLastError();
}

hipError_t launch_quant_mx8(const void* src_bf16, void* dst_fp8, void* dst_scales, int N, int K, int tr, hipStream_t st) {
  if (K % 128 != 0 || (tr != 16 && tr != 8) || (tr == 8 && N % 16 != 0)) return hipErrorInvalidValue;
  const long long total = (long long)((N + 15) / 16) * (K / 128) * 64;
  hipLaunchKernelGGL(quant_mx8_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, (const bf16_t*)src_bf16,
                     (unsigned char*)dst_fp8, (unsigned char*)dst_scales, N, K, tr);
  return hipGetLastError();
}

hipError_t launch_compose(int dtype, const void* A, const void* Bm, const void* bvec, void* out, float* c0, int N, int J, int K,
                          hipStream_t st) {
  const dim3 grid((K + 15) / 16, (N + 15) / 16);
  TW_DISPATCH3(dtype, T, hipLaunchKernelGGL(compose_kernel<T>, grid, dim3(256), 0, st, (const T*)A, (const T*)Bm, (const T*)bvec, (T*)out, c0, N, J, K));
  return hipGetLastError();
}

hipError_t launch_fold_ln(int dtype, void* W, const void* Wsrc, const void* g, const void* beta, const void* bias, float* gw,
                          float* cb, int N, int K, hipStream_t st) {
  dim3 grid((N + 3) / 4);
  TW_DISPATCH3(dtype, T, hipLaunchKernelGGL(fold_ln_kernel<T>, grid, dim3(256), 0, st, (T*)W, (const T*)Wsrc, (const T*)g, (const T*)beta,
                                             (const T*)bias, gw, cb, N, K));
  return hipGetLastError();
}
