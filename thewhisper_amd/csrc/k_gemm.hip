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

#include <cstdint>
#include <cstdlib>
#include <type_traits>

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
  const float e = fmaf(-(p * t), __expf(-az * az), 1.0f);   // (spelled out: one rounding sequence in every instantiation)
  return 0.5f * x * (1.0f + copysignf(e, z));
}

// s_barrier WITHOUT the workgroup-scope fence of __syncthreads(): the fence makes hipcc wait for every outstanding VMEM
// operation (vmcnt(0)) before the barrier, which drains the global->LDS DMA ring and the weight-fragment loads that are
// meant to stay in flight across it.  What the K loops need is ordered by hand: the s_waitcnt vmcnt(n) before the barrier
// covers the DMA data of the tile about to be read, and the LDS reads of the previous tile have been consumed by MFMAs that
// were issued before the barrier.
__device__ __forceinline__ void tw_barrier_only() { asm volatile("s_barrier" ::: "memory"); }

// LDS fragment reads through inline asm.  hipcc cannot tell a ds_read from a preceding global->LDS DMA apart (not even for
// distinct __shared__ arrays): in front of the first consumer of ANY ds_read that follows a DMA in program order it emits
// s_waitcnt vmcnt(0), i.e. it waits for the tile that was requested a moment ago for a LATER iteration - and for the weight
// fragments requested with it.  With compiler-visible reads the K loop therefore has no prefetching at all (measured: 22-25 %
// of the matrix-core peak whatever the tile shape or ring depth).  The asm reads are invisible to that analysis; their
// ordering is done by hand: volatile asm statements keep their program order (barrier, waits, reads), and a fragment becomes
// usable through tw_lds_ready, which waits on lgkmcnt and carries the fragment as an in/out operand so that no consumer can
// be scheduled above it.
typedef __attribute__((address_space(3))) const void* tw_lds_cptr_t;
__device__ __forceinline__ unsigned tw_lds_addr(const void* p) { return (unsigned)(uintptr_t)(tw_lds_cptr_t)p; }
template <int OFF>
__device__ __forceinline__ void tw_lds_read(u32x4_t& dst, unsigned addr) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF));
}
template <int PENDING>
__device__ __forceinline__ void tw_lds_ready(u32x4_t& x) { asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(x) : "n"(PENDING)); }
__device__ __forceinline__ void tw_tie(u32x4_t& x) { asm volatile("" : "+v"(x)); }
// 16-B global load invisible to hipcc's wait-count pass (same reason: with compiler-visible loads in flight next to LDS DMA
// it falls back to vmcnt(0) in front of their first consumer); readiness = the manual s_waitcnt vmcnt + tw_tie
template <int OFF>
__device__ __forceinline__ void tw_gload(u32x4_t& dst, const void* p) {
  asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=v"(dst) : "v"(p), "n"(OFF));
}

// compile-time loops (asm immediates must be constants)
template <int I, int N, typename F> __device__ __forceinline__ void tw_static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    tw_static_for<I + 1, N>(f);
  }
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

template <> struct Vec4<f16_t> {
  f16_t e[4];
  __device__ void load(const f16_t* p) { *reinterpret_cast<uint2*>(e) = *reinterpret_cast<const uint2*>(p); }
  __device__ void store(f16_t* p) const { *reinterpret_cast<uint2*>(p) = *reinterpret_cast<const uint2*>(e); }
  __device__ float get(int i) const { return (float)e[i]; }
  // saturating: HF clamps its float16 hidden states to the largest finite value (HF:models/whisper/modeling_whisper.py:409-411)
  __device__ void set(int i, float f) { e[i] = (f16_t)fminf(fmaxf(f, -65504.f), 65504.f); }
  __device__ f16_t elem(int i) const { return e[i]; }
};

template <typename T>
__device__ __forceinline__ f32x4_t mfma_step(const u32x4_t& wfrag, const u32x4_t& afrag, f32x4_t acc) {
  return tw_mfma32<T>(wfrag, afrag, acc);   // bf16 / f16
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

// Workgroup id -> output tile.  The hardware hands consecutive workgroup ids to the 8 XCDs round-robin (id % 8, MI355X_MICROARCH.md
// "Workgroup dispatch"), and every XCD has its own 4 MiB L2: with a plain 2-D grid the ~64 workgroups resident on one XCD are
// scattered over the whole tile space (QKV projection of 16 x 10 s: 34 different activation tile-rows and all 15 weight
// tile-columns), so every K slice an XCD streams is used by ~3 of its workgroups before it is evicted - the GEMMs ran at the
// rate the L2s MISS at.  swz = 1: the tile list, ordered in groups of GM tile-rows (m fastest inside a group, then n), is cut
// into 8 contiguous chunks and chunk x is walked by XCD x in dispatch order, so the workgroups resident on an XCD at any
// moment cover about GM x 8 neighbouring tiles: each activation slice is shared by ~8 and each weight slice by ~GM of them.
// swz = 0: n fastest over the whole grid (the 2-D grid's order), kept for A/B runs (TW_GEMM_XCD=0).
// Returns false for the padding ids of the last chunk.
__device__ __forceinline__ bool tw_tile_of_block(int id, int Tm, int Tn, int swz, int& tm, int& tn) {
  const int NT = Tm * Tn;
  if (swz == 0) {
    tm = id / Tn;
    tn = id - tm * Tn;
    return id < NT;
  }
  constexpr int GM = 8;
  const int chunk = (NT + 7) >> 3;
  const int local = id >> 3, xcd = id & 7;
  const int t = xcd * chunk + local;
  if (local >= chunk || t >= NT) return false;
  const int per_group = GM * Tn;
  const int g = t / per_group, r = t - g * per_group;
  const int gm = min(GM, Tm - g * GM);
  tn = r / gm;
  tm = g * GM + (r - tn * gm);
  return true;
}

// ---- epilogue shared by the kernels below: a lane holds, per (a,b) MFMA tile, 4 consecutive columns n of row m ----
// Folded pre-LayerNorm, consumer side (GemmEpilogue::stats_in): (mean, rstd) of the BM rows of a block tile into LDS.  Two threads
// per row split the partial statistics the producing GEMM left (parts = K / 32 of them, 8 B each; at most 2 x TW_LN_HALF) and combine
// with one lane swap.  The loads are REQUESTED first in the prologue and the first K tiles next, so the reduction (gemm_ln_publish)
// waits on the oldest requests only; the main loop's first barrier publishes the result to the epilogue.
constexpr int TW_LN_HALF = 20;   // d_model <= 1280 (api.hip falls back to LayerNorm launches above that)
struct LnRaw { f32x2_t v[TW_LN_HALF]; };
template <int BM, int NTHR>
__device__ __forceinline__ void gemm_ln_request(const GemmEpilogue& ep, int m0, int M, int tid, LnRaw& raw) {
  static_assert(2 * BM <= NTHR, "two threads per row of the tile");
  const int parts = ep.stats_in_parts, half = parts >> 1;
  const int m = min(m0 + (tid >> 1), M - 1);
  const f32x2_t* sp = reinterpret_cast<const f32x2_t*>(ep.stats_in) + (long long)m * parts + (tid & 1) * half;
#pragma unroll
  for (int p = 0; p < TW_LN_HALF; ++p) raw.v[p] = sp[min(p, half - 1)];
}
template <int BM, int NTHR>
__device__ __forceinline__ void gemm_ln_publish(const GemmEpilogue& ep, int tid, const LnRaw& raw, f32x2_t* ln_rows) {
  const int half = ep.stats_in_parts >> 1;
  const float inv_k = 1.0f / (float)(ep.stats_in_parts * 32);
  float s = 0.f, ss = 0.f;
#pragma unroll
  for (int p = 0; p < TW_LN_HALF; ++p) {
    s += p < half ? raw.v[p][0] : 0.f;
    ss += p < half ? raw.v[p][1] : 0.f;
  }
  s += __shfl_xor(s, 1);
  ss += __shfl_xor(ss, 1);
  const float mean = s * inv_k;
  const float m2 = mean * mean;
  const float rstd = 1.0f / sqrtf(fmaxf(fmaf(ss, inv_k, -m2), 0.f) + 1e-5f);   // (fused operations spelled out: tw_common.h, tw_ln_scalars)
  const int row = tid >> 1;
  if ((tid & 1) == 0 && row < BM) ln_rows[row] = f32x2_t{mean, rstd};
}

template <typename T, int NT, int MT>
__device__ __forceinline__ void gemm_epilogue(const f32x4_t (&acc)[NT][MT], int m_base, int n_base, int M, int N,
                                              const GemmEpilogue& ep, int fr, int fq, const f32x2_t* ln_rows = nullptr) {
  const T* bias = reinterpret_cast<const T*>(ep.bias);
  const T* res = reinterpret_cast<const T*>(ep.res);
  const int dmodel = ep.H * 64;
  // folded pre-LayerNorm, consumer side (GemmEpilogue): (mean, rstd) of this lane's MT rows were reduced into LDS by gemm_ln_rows
  // while the first K tiles were on their way; the per-column companions are loaded once per column block
  const bool ln = ln_rows != nullptr;
  f32x4_t ln_gw[NT], ln_cb[NT];
  if (ln) {
#pragma unroll
    for (int a = 0; a < NT; ++a) {
      const int n = min(n_base + a * 16 + fq * 4, N - 4);
      ln_gw[a] = *reinterpret_cast<const f32x4_t*>(ep.ln_gw + n);
      ln_cb[a] = *reinterpret_cast<const f32x4_t*>(ep.ln_cb + n);
    }
  }
#pragma unroll
  for (int b = 0; b < MT; ++b) {
    const int m = m_base + b * 16 + fr;
    const bool row_ok = m < M;
    float ln_nmean = 0.f, ln_rstd = 1.f;
    if (ln) {
      const f32x2_t v = ln_rows[b * 16 + fr];
      ln_nmean = -v[0];
      ln_rstd = v[1];
    }
    long long roff = 0;
    if (res && row_ok) roff = rowmap(ep.res_map, ep.res_mod > 0 ? (m % ep.res_mod) : m);
    float st_s[NT], st_ss[NT];   // producer side: this lane's group of four columns of every 16-column tile (tw_stat4)
#pragma unroll
    for (int i = 0; i < NT; ++i) { st_s[i] = 0.f; st_ss[i] = 0.f; }
#pragma unroll
    for (int a = 0; a < NT; ++a) {
      const int n = n_base + a * 16 + fq * 4;
      const bool ok = row_ok && n < N;
      float v[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = acc[a][b][r];
      if (ln) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = fmaf(ln_rstd, fmaf(ln_nmean, ln_gw[a][r], v[r]), ln_cb[a][r]);
      }
      if (bias && ok) {
        Vec4<T> bv;
        bv.load(bias + n);
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] += bv.get(r);
      }
      if (ep.gelu) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = gelu_exact<T>(v[r]);
      }
      if (res && ok) {
        Vec4<T> rv;
        rv.load(res + roff + n);
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] += rv.get(r);
      }
      Vec4<T> ov;
#pragma unroll
      for (int r = 0; r < 4; ++r) ov.set(r, v[r]);
      if (ep.stats_out)    // statistics of the values AS STORED (rounded to T): what the consumer's matrix product sees
        tw_stat4(ok ? ov.get(0) : 0.f, ok ? ov.get(1) : 0.f, ok ? ov.get(2) : 0.f, ok ? ov.get(3) : 0.f, st_s[a], st_ss[a]);
      if (!ok) continue;
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
    if (ep.stats_out) {
      // the 32 columns of a block live in the 4 lanes fq = 0..3 of row fr (8 values each): two lane swaps, lane fq = 0 stores
#pragma unroll
      for (int i = 0; i < NT / 2; ++i) {
        // the fixed tree of tw_stat4's note: the four groups of each tile over the lanes fq, then the two tiles of the block
        const float s = tw_xor32_sum(tw_xor16_sum(st_s[2 * i])) + tw_xor32_sum(tw_xor16_sum(st_s[2 * i + 1]));
        const float ss = tw_xor32_sum(tw_xor16_sum(st_ss[2 * i])) + tw_xor32_sum(tw_xor16_sum(st_ss[2 * i + 1]));
        const int nb = n_base + i * 32;
        if (fq == 0 && row_ok && nb < N)
          reinterpret_cast<f32x2_t*>(ep.stats_out)[(long long)m * (N / 32) + nb / 32] = f32x2_t{s, ss};
      }
    }
  }
}

// Epilogue of the large-M kernel for 16-bit outputs whose rows are CONTIGUOUS over the wavefront's 64 columns (EPI_ROWMAJOR; the q and
// k segments of EPI_QKV_ENC, where the 64 columns are one head), staged through LDS.  Measured with the epilogue compiled out
// (profiles/r04_gemm_parts.txt): it was 40-45 % of every encoder GEMM (fc1 141 -> 77 us, QKV 103 -> 62, out-projection 44 -> 27) - in
// gemm_epilogue a lane owns 4 consecutive columns of one row, so a store instruction writes sixteen 32-byte pieces and a 128-byte line
// of the output is written by four instructions (and the residual read by four).  Here the wavefront writes its float32 values into its
// share of the (idle) activation ring, RB x 16 rows at a time, and reads them back so that 8 lanes cover one row: every store - and the
// residual load - is 16 bytes per lane, 8 full lines per instruction.  The values are staged in float32: bias / LayerNorm fold / GELU
// before, residual add and the ONE rounding to T after, exactly as in gemm_epilogue.  LDS operations of a wavefront execute in order, so
// no barrier is needed between its writes and its reads.  Row stride 68 floats (16-byte aligned, rows 4 banks apart).
constexpr int TW_STAGE_LD = 68;
template <typename T, int NT, int MT, int RB>
__device__ __forceinline__ void gemm_epilogue_staged(const f32x4_t (&acc)[NT][MT], int m_base, int n_base, int M, int N,
                                                     const GemmEpilogue& ep, int fr, int fq, int lane, const f32x2_t* ln_rows,
                                                     float* stage) {
  static_assert(NT == 4 && sizeof(T) == 2, "64 columns of 16-bit elements per wavefront");
  const T* bias = reinterpret_cast<const T*>(ep.bias);
  const T* res = reinterpret_cast<const T*>(ep.res);
  const bool ln = ln_rows != nullptr;
  f32x4_t ln_gw[NT], ln_cb[NT], bv[NT];
#pragma unroll
  for (int a = 0; a < NT; ++a) {
    const int n = n_base + a * 16 + fq * 4;
    bv[a] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    if (ln) {
      ln_gw[a] = *reinterpret_cast<const f32x4_t*>(ep.ln_gw + n);
      ln_cb[a] = *reinterpret_cast<const f32x4_t*>(ep.ln_cb + n);
    } else if (bias) {
      Vec4<T> b4;
      b4.load(bias + n);
#pragma unroll
      for (int r = 0; r < 4; ++r) bv[a][r] = b4.get(r);
    }
  }
  // where this wavefront's rows go: EPI_ROWMAJOR out + cmap(m) + n; EPI_QKV_ENC (segment 0 / 1) ((b*H + h)*T + t)*64 + dd
  const int dmodel = ep.H * 64;
  const int seg = ep.mode == EPI_QKV_ENC ? n_base / dmodel : 0;
  const int head = ep.mode == EPI_QKV_ENC ? (n_base - seg * dmodel) >> 6 : 0;
  T* const obase = reinterpret_cast<T*>(seg == 0 ? ep.out : ep.out2);
  const int prow = lane >> 3, pcol = (lane & 7) * 8;   // read-back role: row prow (+ 8 i) of the chunk, columns pcol .. pcol + 7
#pragma unroll
  for (int b0 = 0; b0 < MT; b0 += RB) {
    constexpr int dummy = 0; (void)dummy;
    const int nb = (MT - b0) < RB ? (MT - b0) : RB;     // 16-row tiles in this chunk (compile-time after unrolling)
#pragma unroll
    for (int bb = 0; bb < RB; ++bb) {
      if (bb >= nb) break;
      const int b = b0 + bb;
      float ln_nmean = 0.f, ln_rstd = 1.f;
      if (ln) {
        const f32x2_t v = ln_rows[b * 16 + fr];
        ln_nmean = -v[0];
        ln_rstd = v[1];
      }
#pragma unroll
      for (int a = 0; a < NT; ++a) {
        f32x4_t v = acc[a][b];
        if (ln) {
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = fmaf(ln_rstd, fmaf(ln_nmean, ln_gw[a][r], v[r]), ln_cb[a][r]);
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] += bv[a][r];
        }
        if (ep.gelu) {
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = gelu_exact<T>(v[r]);
        }
        *reinterpret_cast<f32x4_t*>(stage + (bb * 16 + fr) * TW_STAGE_LD + a * 16 + fq * 4) = v;
      }
    }
#pragma unroll
    for (int i = 0; i < 2 * RB; ++i) {
      if (i >= 2 * nb) break;
      const int row = i * 8 + prow;
      const int m = m_base + b0 * 16 + row;
      const bool ok = m < M;
      const f32x4_t lo = *reinterpret_cast<const f32x4_t*>(stage + row * TW_STAGE_LD + pcol);
      const f32x4_t hi = *reinterpret_cast<const f32x4_t*>(stage + row * TW_STAGE_LD + pcol + 4);
      float v[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      if (res && ok) {
        const long long roff = rowmap(ep.res_map, ep.res_mod > 0 ? (m % ep.res_mod) : m);
        const uint4 rq = *reinterpret_cast<const uint4*>(res + roff + n_base + pcol);
        const T* rt = reinterpret_cast<const T*>(&rq);
#pragma unroll
        for (int r = 0; r < 8; ++r) v[r] += (float)rt[r];
      }
      Vec4<T> o0, o1;
#pragma unroll
      for (int r = 0; r < 4; ++r) { o0.set(r, v[r]); o1.set(r, v[4 + r]); }
      if (ep.stats_out) {   // (sum, sum of squares) of the values AS STORED, per row and 32-column block: 4 lanes x 8 columns
        float sa, ssa, sb, ssb;   // groups 2j and 2j + 1 of the block (tw_stat4's note: the same tree as the plain epilogue)
        tw_stat4(ok ? o0.get(0) : 0.f, ok ? o0.get(1) : 0.f, ok ? o0.get(2) : 0.f, ok ? o0.get(3) : 0.f, sa, ssa);
        tw_stat4(ok ? o1.get(0) : 0.f, ok ? o1.get(1) : 0.f, ok ? o1.get(2) : 0.f, ok ? o1.get(3) : 0.f, sb, ssb);
        float s = sa + sb, ss = ssa + ssb;
        s += __shfl_xor(s, 1);
        ss += __shfl_xor(ss, 1);
        s += __shfl_xor(s, 2);
        ss += __shfl_xor(ss, 2);
        if ((lane & 3) == 0 && ok)
          reinterpret_cast<f32x2_t*>(ep.stats_out)[(long long)m * (N / 32) + (n_base + pcol) / 32] = f32x2_t{s, ss};
      }
      if (!ok) continue;
      T* dst;
      if (ep.mode == EPI_ROWMAJOR) {
        dst = obase + rowmap(ep.c_map, m) + n_base + pcol;
      } else {
        const int bidx = m / ep.T, t = m - bidx * ep.T;
        dst = obase + ((long long)(bidx * ep.H + head) * ep.T + t) * 64 + pcol;
      }
      uint4 q;
      *reinterpret_cast<uint2*>(&q.x) = *reinterpret_cast<const uint2*>(o0.e);
      *reinterpret_cast<uint2*>(&q.z) = *reinterpret_cast<const uint2*>(o1.e);
      *reinterpret_cast<uint4*>(dst) = q;
    }
  }
}

// The V segment of EPI_QKV_ENC through the same staging: the output is V^T ([stream, head][64 dims][Tp keys]), so the wavefront writes
// its values TRANSPOSED into LDS ([dim][token of the chunk], conflict-free: 16 tokens x 4 dims apart by 16 banks) and every lane reads
// 4 consecutive tokens of one dim back: 8-byte stores, 8 lanes = 64 contiguous bytes of a V^T row, instead of one 2-byte store per
// element.  4 consecutive tokens never straddle two clips (the caller guarantees T % 4 == 0 and the chunk starts on a multiple of 16).
template <typename T, int NT, int MT, int RB>
__device__ __forceinline__ void gemm_epilogue_staged_vt(const f32x4_t (&acc)[NT][MT], int m_base, int n_base, int M, int N,
                                                        const GemmEpilogue& ep, int fr, int fq, int lane, const f32x2_t* ln_rows,
                                                        float* stage) {
  static_assert(NT == 4 && sizeof(T) == 2, "64 columns of 16-bit elements per wavefront");
  constexpr int LD = RB * 16 + 4;     // floats per dim row of the transposed chunk
  const T* bias = reinterpret_cast<const T*>(ep.bias);
  const bool ln = ln_rows != nullptr;
  f32x4_t ln_gw[NT], ln_cb[NT], bv[NT];
#pragma unroll
  for (int a = 0; a < NT; ++a) {
    const int n = n_base + a * 16 + fq * 4;
    bv[a] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    if (ln) {
      ln_gw[a] = *reinterpret_cast<const f32x4_t*>(ep.ln_gw + n);
      ln_cb[a] = *reinterpret_cast<const f32x4_t*>(ep.ln_cb + n);
    } else if (bias) {
      Vec4<T> b4;
      b4.load(bias + n);
#pragma unroll
      for (int r = 0; r < 4; ++r) bv[a][r] = b4.get(r);
    }
  }
  const int dmodel = ep.H * 64;
  const int head = (n_base - 2 * dmodel) >> 6;
  T* const obase = reinterpret_cast<T*>(ep.out3);
#pragma unroll
  for (int b0 = 0; b0 < MT; b0 += RB) {
    const int nb = (MT - b0) < RB ? (MT - b0) : RB;
#pragma unroll
    for (int bb = 0; bb < RB; ++bb) {
      if (bb >= nb) break;
      const int b = b0 + bb;
      float ln_nmean = 0.f, ln_rstd = 1.f;
      if (ln) {
        const f32x2_t v = ln_rows[b * 16 + fr];
        ln_nmean = -v[0];
        ln_rstd = v[1];
      }
#pragma unroll
      for (int a = 0; a < NT; ++a) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float v = acc[a][b][r];
          v = ln ? fmaf(ln_rstd, fmaf(ln_nmean, ln_gw[a][r], v), ln_cb[a][r]) : v + bv[a][r];
          stage[(a * 16 + fq * 4 + r) * LD + bb * 16 + fr] = v;
        }
      }
    }
    const int G = nb * 4;                 // groups of 4 tokens per dim in this chunk
    for (int task = lane; task < 64 * G; task += 64) {
      const int dd = task / G, g = task - dd * G;
      const f32x4_t v = *reinterpret_cast<const f32x4_t*>(stage + dd * LD + g * 4);
      const int m = m_base + b0 * 16 + g * 4;
      if (m >= M) continue;               // (M % 4 == 0: the 4 tokens are in or out together)
      const int bidx = m / ep.T, t = m - bidx * ep.T;
      Vec4<T> o;
#pragma unroll
      for (int r = 0; r < 4; ++r) o.set(r, v[r]);
      *reinterpret_cast<uint2*>(obase + ((long long)(bidx * ep.H + head) * 64 + dd) * ep.Tp + t) = *reinterpret_cast<const uint2*>(o.e);
    }
  }
}

// Epilogue of the cross-K/V projection in TW_BF16_MXFP8 contexts: the wavefront's 64 columns are exactly one head of K or of
// V, so the per-key maximum over the head is 16 in-lane values and two lane swaps; every key gets one power-of-two scale
// byte (sb = max(E - 7, 1), E = biased exponent of the maximum: the rule of sk_quant_mx8 in k_decode.hip) and its 64 values
// are stored as e4m3(bf16(v) / 2^(sb-127)) in the fp8 fragment layouts of tw_common.h.
typedef short tw_s16x2 __attribute__((ext_vector_type(2)));
template <typename T, int NT, int MT>
__device__ __forceinline__ void gemm_epilogue_kv8(const f32x4_t (&acc)[NT][MT], int m_base, int n_base, int M, int N,
                                                  const GemmEpilogue& ep, int fr, int fq) {
  static_assert(NT == 4, "one 64-column head per wavefront");
  if constexpr (ElemTraits<T>::kCode == 1) {   // fp8 cross K / V caches exist in bf16 contexts only
    const T* bias = reinterpret_cast<const T*>(ep.bias);
    const int dmodel = ep.H * 64;
    if (n_base >= N) return;
    const int seg = n_base / dmodel;                 // 0: K, 1: V
    const int h = (n_base - seg * dmodel) >> 6;
    float bv[NT][4];
#pragma unroll
    for (int a = 0; a < NT; ++a) {
      Vec4<T> b4;
      if (bias) b4.load(bias + n_base + a * 16 + fq * 4);
#pragma unroll
      for (int r = 0; r < 4; ++r) bv[a][r] = bias ? b4.get(r) : 0.f;
    }
#pragma unroll
    for (int b = 0; b < MT; ++b) {
      const int m = m_base + b * 16 + fr;
      const int mc = min(m, M - 1);
      const int bidx = mc / ep.T, t = mc % ep.T;
      bf16x2_t pk[NT][2];
      unsigned amax = 0;
#pragma unroll
      for (int a = 0; a < NT; ++a) {
        pk[a][0] = bf16x2_t{(bf16_t)(acc[a][b][0] + bv[a][0]), (bf16_t)(acc[a][b][1] + bv[a][1])};
        pk[a][1] = bf16x2_t{(bf16_t)(acc[a][b][2] + bv[a][2]), (bf16_t)(acc[a][b][3] + bv[a][3])};
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const unsigned u = __builtin_bit_cast(unsigned, pk[a][j]) & 0x7fff7fffu;
          amax = max(amax, max(u & 0xffffu, u >> 16));
        }
      }
      // the 64 columns of row m live in the 4 lanes fr, fr+16, fr+32, fr+48
      float am = __builtin_bit_cast(float, amax << 16);   // the bf16 pattern as the float32 it denotes (a normal number)
      am = tw_xor32_max(tw_xor16_max(am));
      amax = __builtin_bit_cast(unsigned, am) >> 16;
      const int sb = max((int)(amax >> 7) - 7, 1);
      const float X = __builtin_bit_cast(float, (unsigned)sb << 23);
      if (m < M) {
        const long long hb = (long long)(bidx * ep.H + h) * ep.Tp;       // (stream, head) slab: Tp*64 bytes, Tp scale bytes
        unsigned char* dst = reinterpret_cast<unsigned char*>(seg == 0 ? ep.out : ep.out2) + hb * 64;
#pragma unroll
        for (int a = 0; a < NT; ++a) {
          tw_s16x2 q = {0, 0};
          q = __builtin_amdgcn_cvt_scalef32_pk_fp8_bf16(q, pk[a][0], X, false);
          q = __builtin_amdgcn_cvt_scalef32_pk_fp8_bf16(q, pk[a][1], X, true);
          const unsigned w = __builtin_bit_cast(unsigned, q);
          const int c = a * 16 + fq * 4;
          if (seg == 0) {
            *reinterpret_cast<unsigned*>(dst + tw_kf8_index(t, c)) = w;     // 4 dims of one key: 4 consecutive bytes
          } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) dst[tw_vtf8_index(t, c + r)] = (unsigned char)(w >> (8 * r));
          }
        }
        // K scales are stored [64-key group][key % 16][key tile] so that a lane of the attention kernel (one key row of each
        // of the 4 tiles) fetches its four scale bytes with ONE 32-bit load; V scales in key order (lane = key there)
        if (fq == 0) {
          if (seg == 0) reinterpret_cast<unsigned char*>(ep.out3)[hb + (t & ~63) + (t & 15) * 4 + ((t >> 4) & 3)] = (unsigned char)sb;
          else reinterpret_cast<unsigned char*>(ep.out4)[hb + t] = (unsigned char)sb;
        }
      }
    }
  }
}

// The weight operand W[N,K] of every GEMM is stored FRAGMENT-MAJOR (launch_tile_weights with 16-row tiles, done once at
// tw_finalize_weights): for n-tile t = n/16 and k-step s = k/(4E) one contiguous 1-KiB block holds the MFMA A operand exactly
// as the 64 lanes consume it (lane = kq*16 + n%16 holds W[n][s*4E + kq*E .. +E]); element offset ((t*S + s)*64 + lane)*E with
// S = K/(4E).  Either kernel moves whole 1-KiB fragments: kernel 1 by DMA into LDS (fragment reads are then linear and
// conflict-free), kernel 2 straight into registers.

// Kernel 1.  BM x BN block tile, WM x WN wavefronts (each owns a (BM/WM) x (BN/WN) sub-tile), both operands through an
// ST-stage LDS ring filled by global->LDS DMA with prefetch distance ST-1.  One barrier per K tile: after it, tile kt is
// visible to every wavefront and buffer (kt-1) % ST - consumed in the previous iteration - is free for tile kt + ST - 1.
// Used for small M (single stream) where the grid needs small tiles.
template <typename T, int BM, int BN, int WM, int WN, int ST>
__global__ __launch_bounds__(WM * WN * 64) void gemm_kernel(const T* __restrict__ A, RowMap amap, const T* __restrict__ W,
                                                              int M, int N, int K, int Tm, int Tn, int swz, GemmEpilogue ep) {
  constexpr int E = ElemTraits<T>::kPer16B;  // elements per 16-B vector
  constexpr int BKE = 8 * E;                 // elements per K tile (two MFMA k-steps)
  constexpr int NWAVE = WM * WN;
  constexpr int NTHR = NWAVE * 64;
  constexpr int AV = BM * 8 / NTHR;          // 16-B vectors per thread per tile (activations)
  constexpr int WV = BN * 8 / NTHR;          // (weights): BN/8 fragments of 64 vectors per K tile
  constexpr int RM = BM / WM, RN = BN / WN;  // rows / columns of a wavefront's sub-tile
  constexpr int MT = RM / 16;                // 16-row MFMA tiles per wave along M
  constexpr int NT = RN / 16;                // along N
  static_assert(AV >= 1 && WV >= 1 && MT >= 1 && NT >= 1, "tile / wavefront layout");
  constexpr int STAGE = 8 * (BM + BN);       // 16-B vectors per ring stage
  constexpr int PER = AV + WV;               // DMA requests per thread and stage
  static_assert(ST >= 2 && (ST - 2) * PER <= 63, "ring depth against the 6-bit vmcnt");
  // the ring lives in dynamic LDS (any depth the 160 KB of a CU and the 6-bit vmcnt allow; deep rings were tried, see gemm_dispatch)
  extern __shared__ __attribute__((aligned(16))) unsigned char gemm_smem[];
  u32x4_t* lds = reinterpret_cast<u32x4_t*>(gemm_smem);
  f32x2_t* ln_rows = reinterpret_cast<f32x2_t*>(gemm_smem + (size_t)ST * STAGE * 16);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wn = wave % WN;  // wave position along N
  const int wm = wave / WN;  // along M
  int tile_m, tile_n;
  if (!tw_tile_of_block(blockIdx.x, Tm, Tn, swz, tile_m, tile_n)) return;   // (whole workgroup: before any barrier)
  const int n0 = tile_n * BN;
  const int m0 = tile_m * BM;
  const int S = K / (4 * E);

  // Activations: the DMA writes the 64 lanes of a wavefront to 64 consecutive 16-B LDS slots, so the swizzle is applied on
  // the SOURCE side: linear LDS position p (16-B units) of a tile holds row p>>3, k-slot (p&7) ^ ((row>>1)&7); 8 consecutive
  // lanes still cover one 128-B row segment of global memory (coalesced), and the fragment read - 16 lanes = 16 consecutive
  // rows of one k-slot, ds_read_b128 - touches every bank exactly once.
  const T* asrc[AV];
#pragma unroll
  for (int i = 0; i < AV; ++i) {
    const int p = (i * NWAVE + wave) * 64 + lane;
    const int row = p >> 3, slot = (p & 7) ^ ((row >> 1) & 7);
    int m = m0 + row;
    if (m >= M) m = M - 1;
    asrc[i] = A + rowmap(amap, m) + slot * E;
  }
  // Weights: DMA piece q = i*NWAVE + wave is fragment (n-tile q>>1, k-step q&1) of this K tile
  const T* wsrc[WV];
#pragma unroll
  for (int i = 0; i < WV; ++i) {
    const int q = i * NWAVE + wave;
    const int tile = min(n0 / 16 + (q >> 1), N / 16 - 1);
    wsrc[i] = W + (((long long)tile * S + (q & 1)) * 64 + lane) * E;
  }
  typedef __attribute__((address_space(1))) const void* gptr_t;
  typedef __attribute__((address_space(3))) void* lptr_t;
  auto issue_tile = [&](int kt, int b) {
    u32x4_t* sa = lds + b * STAGE;
    u32x4_t* sw = sa + 8 * BM;
#pragma unroll
    for (int i = 0; i < AV; ++i)
      __builtin_amdgcn_global_load_lds((gptr_t)(asrc[i] + kt * BKE), (lptr_t)(sa + (i * NWAVE + wave) * 64), 16, 0, 0);
#pragma unroll
    for (int i = 0; i < WV; ++i)
      __builtin_amdgcn_global_load_lds((gptr_t)(wsrc[i] + (long long)kt * (2 * 64 * E)), (lptr_t)(sw + (i * NWAVE + wave) * 64), 16, 0, 0);
  };

  f32x4_t acc[NT][MT];
#pragma unroll
  for (int a = 0; a < NT; ++a)
#pragma unroll
    for (int b = 0; b < MT; ++b) acc[a][b] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  const int nk = K / BKE;
  const int fr = lane & 15;  // fragment row within a 16-row tile
  const int fq = lane >> 4;  // k-slot quad
  unsigned a_addr[2];
#pragma unroll
  for (int kk = 0; kk < 2; ++kk) {
    const int row = wm * RM + fr;
    a_addr[kk] = tw_lds_addr(lds) + (unsigned)(row * 8 + ((kk * 4 + fq) ^ ((row >> 1) & 7))) * 16;
  }
  const unsigned w_addr = tw_lds_addr(lds) + (unsigned)(8 * BM + ((wn * RN) / 16) * 2 * 64 + lane) * 16;
  LnRaw ln_raw;
  if (ep.stats_in) gemm_ln_request<BM, NTHR>(ep, m0, M, tid, ln_raw);
#pragma unroll
  for (int t = 0; t < ST - 1; ++t)
    if (t < nk) issue_tile(t, t);
  if (ep.stats_in) gemm_ln_publish<BM, NTHR>(ep, tid, ln_raw, ln_rows);
  int buf = 0;
  for (int kt = 0; kt < nk; ++kt) {
    // tile kt has landed: this wavefront's own DMA by vmcnt (requests retire in order; the ST-2 younger tiles may stay in
    // flight), the other wavefronts' by the barrier
    if (ST > 2 && kt + ST - 2 < nk) {
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"((ST - 2) * PER) : "memory");
    } else if (ST > 3) {
      // the last ST - 2 tiles: r = nk - 1 - kt younger tiles are in flight (r < ST - 2)
      const int r = nk - 1 - kt;
      tw_static_for<0, (ST > 3 ? ST - 2 : 1)>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        if (r == i) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(i * PER) : "memory");
      });
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    tw_barrier_only();
    if (kt + ST - 1 < nk) {
      int nb = buf + ST - 1;
      if (nb >= ST) nb -= ST;
      issue_tile(kt + ST - 1, nb);
    }
    const unsigned stage_off = (unsigned)buf * (STAGE * 16);
    tw_static_for<0, 2>([&](auto kkc) {
      constexpr int kk = decltype(kkc)::value;
      u32x4_t af[MT], wf[NT];
      // activation fragment b: row wm*RM + b*16 + fr (the swizzle term (row>>1)&7 does not depend on b), k-slot kk*4 + fq
      tw_static_for<0, MT>([&](auto bc) { tw_lds_read<decltype(bc)::value * 2048>(af[decltype(bc)::value], a_addr[kk] + stage_off); });
      // weight fragment a of this wavefront: linear, 1 KiB per (n-tile, k-step)
      tw_static_for<0, NT>([&](auto ac) { tw_lds_read<(decltype(ac)::value * 2 + kk) * 1024>(wf[decltype(ac)::value], w_addr + stage_off); });
      tw_lds_ready<0>(af[0]);
#pragma unroll
      for (int b = 1; b < MT; ++b) tw_tie(af[b]);
#pragma unroll
      for (int a = 0; a < NT; ++a) tw_tie(wf[a]);
#pragma unroll
      for (int a = 0; a < NT; ++a)
#pragma unroll
        for (int b = 0; b < MT; ++b) acc[a][b] = mfma_step<T>(wf[a], af[b], acc[a][b]);
    });
    if (++buf == ST) buf = 0;
  }
  gemm_epilogue<T, NT, MT>(acc, m0 + wm * RM, n0 + wn * RN, M, N, ep, fr, fq, ep.stats_in ? ln_rows + wm * RM : nullptr);
}

// Kernel 2 (large M: the batched encoder, the cross-K/V projection).  The round-1 kernel staged both operands through LDS in
// 128 x 128 tiles and ran at 22 % of the matrix-core peak: per K tile the LDS pipe had to take 32 KB of DMA writes and serve
// 64 fragment reads for 128 MFMAs, about as many LDS cycles as MFMA cycles, and 32 KB per 2.1 MFLOP is also right at what a CU
// can pull from L2 (64 B/clk).  Here only the ACTIVATION tile goes through LDS; every wavefront owns a BM x 64 slab of the
// block tile (all NW wavefronts side by side along N, none stacked along M) and takes its weight fragments - which nobody
// else in the workgroup needs - straight from global memory into registers, 1 KiB contiguous per request, one K tile ahead.
// Per K tile and CU: 16 KB of DMA + 32 fragment reads per wavefront for 64 MFMAs each, and (16 + 32) KB for 4.2 MFLOP.
template <typename T, int BM, int NW, int ST>
__global__ __launch_bounds__(NW * 64, 2) void gemm_wreg_kernel(const T* __restrict__ A, RowMap amap, const T* __restrict__ W,
                                                             int M, int N, int K, int Tm, int Tn, int swz, GemmEpilogue ep) {
  constexpr int E = ElemTraits<T>::kPer16B;
  constexpr int BKE = 8 * E;
  constexpr int BN = NW * 64;
  constexpr int NTHR = NW * 64;
  // DMA pieces (64 x 16 B, one per wavefront request) per thread and stage.  BM need not be a multiple of NTHR / 8 rows (80, 96,
  // 112: tile heights that divide the grid better, see gemm_pick_bm): the last piece then also fetches rows past the tile
  // (clamped to the matrix; they land in LDS rows nobody reads)
  constexpr int AV = (BM * 8 + NTHR - 1) / NTHR;
  constexpr int MT = BM / 16, NT = 4;
  static_assert(BM % 16 == 0 && AV >= 1 && (ST == 2 || ST == 3), "tile layout");
  constexpr int STAGE = AV * NTHR;
  __shared__ u32x4_t lds[ST * STAGE];
  __shared__ f32x2_t ln_rows[BM];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  int tile_m, tile_n;
  if (!tw_tile_of_block(blockIdx.x, Tm, Tn, swz, tile_m, tile_n)) return;   // (whole workgroup: before any barrier)
  const int n0 = tile_n * BN + wave * 64;
  const int m0 = tile_m * BM;
  const int S = K / (4 * E);
  const int fr = lane & 15, fq = lane >> 4;

  const T* asrc[AV];
#pragma unroll
  for (int i = 0; i < AV; ++i) {
    const int p = (i * NW + wave) * 64 + lane;
    const int row = p >> 3, slot = (p & 7) ^ ((row >> 1) & 7);
    int m = m0 + row;
    if (m >= M) m = M - 1;
    asrc[i] = A + rowmap(amap, m) + slot * E;
  }
  const T* wsrc[NT];   // this wavefront's 4 n-tiles, k-step 0
#pragma unroll
  for (int a = 0; a < NT; ++a) wsrc[a] = W + ((long long)min(n0 / 16 + a, N / 16 - 1) * S * 64 + lane) * E;
  typedef __attribute__((address_space(1))) const void* gptr_t;
  typedef __attribute__((address_space(3))) void* lptr_t;
  auto issue_a = [&](int kt, int b) {
    u32x4_t* sa = lds + b * STAGE;
#pragma unroll
    for (int i = 0; i < AV; ++i)
      __builtin_amdgcn_global_load_lds((gptr_t)(asrc[i] + kt * BKE), (lptr_t)(sa + (i * NW + wave) * 64), 16, 0, 0);
  };
  // two register sets for the weight fragments, used alternately (the K loop is unrolled by two): a copy "current <- next"
  // would let the compiler sink the copy - and with it the wait for the freshly issued loads - into the MFMA block
  u32x4_t w0[NT][2], w1[NT][2];
  auto load_w = [&](u32x4_t (&w)[NT][2], int kt) {
#pragma unroll
    for (int a = 0; a < NT; ++a) {
      const T* p = wsrc[a] + (long long)kt * (2 * 64 * E);
      tw_gload<0>(w[a][0], p);
      tw_gload<1024>(w[a][1], p);
    }
  };

  f32x4_t acc[NT][MT];
  const int nk = K / BKE;
  int buf = 0;
  unsigned a_addr[2];
#pragma unroll
  for (int kk = 0; kk < 2; ++kk) a_addr[kk] = tw_lds_addr(lds) + (unsigned)(fr * 8 + ((kk * 4 + fq) ^ ((fr >> 1) & 7))) * 16;
  // one K tile: A(kt) and `cur` = W(kt) have landed (vmcnt retires in order: with three stages the younger A(kt+1) may stay
  // in flight); then W(kt+1) -> `nxt` and A(kt+ST-1) are requested, in that order, and tile kt is multiplied
  auto step = [&](int kt, u32x4_t (&cur)[NT][2], u32x4_t (&nxt)[NT][2]) {
    // (requests are unconditional - past the last tile they re-fetch tile nk-1 into a free stage / the idle register set -
    // so that the loop body is straight-line code: with requests inside branches hipcc's wait-count bookkeeping gives up
    // at the join and drains everything, vmcnt(0), in front of the first MFMA)
    if (ST == 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(AV) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int a = 0; a < NT; ++a) { tw_tie(cur[a][0]); tw_tie(cur[a][1]); }
    tw_barrier_only();
    load_w(nxt, min(kt + 1, nk - 1));
    {
      int nb = buf + ST - 1;
      if (nb >= ST) nb -= ST;
      issue_a(min(kt + ST - 1, nk - 1), nb);
    }
    const unsigned stage_off = (unsigned)buf * (STAGE * 16);
    tw_static_for<0, 2>([&](auto kkc) {
      constexpr int kk = decltype(kkc)::value;
      u32x4_t af[MT];
      tw_static_for<0, MT>([&](auto bc) { tw_lds_read<decltype(bc)::value * 2048>(af[decltype(bc)::value], a_addr[kk] + stage_off); });
      // fragments are consumed in arrival order: MFMAs on af[b] start while af[b+1..] are still on their way
      tw_static_for<0, MT>([&](auto bc) {
        constexpr int b = decltype(bc)::value;
        tw_lds_ready<MT - 1 - b>(af[b]);
#pragma unroll
        for (int a = 0; a < NT; ++a) acc[a][b] = mfma_step<T>(cur[a][kk], af[b], acc[a][b]);
      });
    });
    if (++buf == ST) buf = 0;
  };
  // request order: [row statistics of a folded LayerNorm,] A(0), W(0), A(1) | per tile: W(kt+1), A(kt+ST-1)
  LnRaw ln_raw;
  if (ep.stats_in) gemm_ln_request<BM, NTHR>(ep, m0, M, tid, ln_raw);
  issue_a(0, 0);
  load_w(w0, 0);
  if (ST == 3) issue_a(min(1, nk - 1), 1);
  if (ep.stats_in) gemm_ln_publish<BM, NTHR>(ep, tid, ln_raw, ln_rows);
#pragma unroll
  for (int a = 0; a < NT; ++a)   // (zeroed here, after the prologue's registers are free again)
#pragma unroll
    for (int b = 0; b < MT; ++b) acc[a][b] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  for (int kt = 0; kt < nk; kt += 2) {
    step(kt, w0, w1);
    if (kt + 1 < nk) step(kt + 1, w1, w0);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the redundant tail requests must not outlive the LDS allocation
  if constexpr (sizeof(T) == 2) {
    // rows contiguous over this wavefront's 64 columns: the epilogue staged through this wavefront's share of the ring
    constexpr int WAVE_FLOATS = ST * STAGE * 4 / NW;
    constexpr int RB = WAVE_FLOATS >= 2 * 16 * TW_STAGE_LD ? 2 : 1;
    static_assert(WAVE_FLOATS >= 16 * TW_STAGE_LD, "a 16-row chunk per wavefront");
    const bool rows64 = n0 + 64 <= N && (ep.mode == EPI_ROWMAJOR || (ep.mode == EPI_QKV_ENC && n0 < 2 * ep.H * 64));
    if (ep.staged && rows64) {   // (wavefront-uniform; the barrier below is reached by every wavefront: ep.staged is a launch constant)
      tw_barrier_only();         // every wavefront has read its last fragments and every DMA has landed: the ring is free
      gemm_epilogue_staged<T, NT, MT, RB>(acc, m0, n0, M, N, ep, fr, fq, lane, ep.stats_in ? ln_rows : nullptr,
                                          reinterpret_cast<float*>(lds) + wave * WAVE_FLOATS);
      return;
    }
    if (ep.staged) tw_barrier_only();   // (keep the barrier count equal across the workgroup)
    constexpr int RBT = WAVE_FLOATS >= 64 * (2 * 16 + 4) ? 2 : 1;
    static_assert(WAVE_FLOATS >= 64 * (16 + 4), "a transposed 16-token chunk per wavefront");
    if (ep.staged == 1 && ep.mode == EPI_QKV_ENC && n0 >= 2 * ep.H * 64 && n0 + 64 <= N && ep.T % 4 == 0 && M % 4 == 0) {   // the V segment
      gemm_epilogue_staged_vt<T, NT, MT, RBT>(acc, m0, n0, M, N, ep, fr, fq, lane, ep.stats_in ? ln_rows : nullptr,
                                              reinterpret_cast<float*>(lds) + wave * WAVE_FLOATS);
      return;
    }
  }
  if (ep.mode == EPI_KV_CROSS8) gemm_epilogue_kv8<T, NT, MT>(acc, m0, n0, M, N, ep, fr, fq);
  else gemm_epilogue<T, NT, MT>(acc, m0, n0, M, N, ep, fr, fq, ep.stats_in ? ln_rows : nullptr);
}

}  // namespace

static int gemm_env(const char* name, int dflt) {
  const char* e = getenv(name);
  return e ? atoi(e) : dflt;
}

// 1-D grid of 8 * ceil(tiles / 8) workgroups (tw_tile_of_block).  TW_GEMM_XCD: 1 (default) = XCD-aware order for the large-M kernel
// only, 0 = row-major everywhere, 2 = XCD-aware for both kernels.  Measured (profiles/r04_encoder_xcd_ab.txt, GPU to itself):
// encoder of 16 x 10 / 15 / 30 s 16.90 -> 16.51 / 24.03 -> 22.72 / 51.55 -> 49.10 ms; the small-M kernel (one stream: 80-320 small
// tiles) loses 1.5-5 % with it (3.56 -> 3.75 ms at 10 s), hence off there.
static int gemm_grid(int Tm, int Tn, bool large, int* swz) {
  static const int xcd = gemm_env("TW_GEMM_XCD", 1);
  const bool on = xcd == 2 || (xcd == 1 && large);
  *swz = on ? 1 : 0;
  const int NT = Tm * Tn;
  return on ? 8 * ((NT + 7) / 8) : NT;
}

template <typename T, int BM, int BN, int WM, int WN, int ST>
static hipError_t gemm_go(const void* A, RowMap amap, const void* W, int M, int N, int K, const GemmEpilogue& ep, hipStream_t st) {
  const int Tn = (N + BN - 1) / BN, Tm = (M + BM - 1) / BM;
  int swz;
  dim3 grid(gemm_grid(Tm, Tn, false, &swz));
  constexpr size_t lds_bytes = (size_t)ST * 8 * (BM + BN) * 16 + (size_t)BM * sizeof(f32x2_t);   // ring + folded-LayerNorm rows
  static_assert(lds_bytes <= 160 * 1024, "LDS of a compute unit");
  static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_kernel<T, BM, BN, WM, WN, ST>),
                                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
  if (attr != hipSuccess) return attr;
  hipLaunchKernelGGL((gemm_kernel<T, BM, BN, WM, WN, ST>), grid, dim3(WM * WN * 64), lds_bytes, st, reinterpret_cast<const T*>(A), amap,
                     reinterpret_cast<const T*>(W), M, N, K, Tm, Tn, swz, ep);
  return hipGetLastError();
}

template <typename T, int BM, int NW, int ST>
static hipError_t gemm_wreg_go(const void* A, RowMap amap, const void* W, int M, int N, int K, const GemmEpilogue& ep, hipStream_t st) {
  const int Tn = (N + NW * 64 - 1) / (NW * 64), Tm = (M + BM - 1) / BM;
  int swz;
  dim3 grid(gemm_grid(Tm, Tn, true, &swz));
  hipLaunchKernelGGL((gemm_wreg_kernel<T, BM, NW, ST>), grid, dim3(NW * 64), 0, st, reinterpret_cast<const T*>(A), amap,
                     reinterpret_cast<const T*>(W), M, N, K, Tm, Tn, swz, ep);
  return hipGetLastError();
}

template <typename T>
static hipError_t gemm_dispatch(const void* A, RowMap amap, const void* W, int M, int N, int K,
                                const GemmEpilogue& ep, hipStream_t st) {
  constexpr int E = ElemTraits<T>::kPer16B;
  if (M <= 0) return hipSuccess;
  if (K % (8 * E) != 0 || N % 64 != 0) return hipErrorInvalidValue;
  // Tile choice (TW_GEMM_CFG forces one for experiments):
  //   5: 128 x 256, 4 wavefronts side by side, weights straight to registers, 3-stage activation ring (kernel 2)  - large M
  //   6: same, 2 stages;   8: 64 x 256 (narrow N at large M: twice the workgroups)
  //   4: 128 x 128, 4 wavefronts, both operands through a 2-stage LDS ring (the round-1 shape)
  //   1: 128 x 64 (3 stages);  12: 64 x 64 (4 stages);  0: 64 x 64 (2 stages)  - small M (one to three streams)
  static const int forced = gemm_env("TW_GEMM_CFG", -1);
  static const int narrow = gemm_env("TW_GEMM_NARROW", 5);   // tile config for N <= 2048 at large M (experiments)
  // kernel 2 from this many 128 x 128 tiles on: 128 x 256 tiles need >= 256 workgroups to cover the chip.  One 30 s chunk's fc1
  // (M = 1500, N = 5120: 480 tiles = 240 workgroups) is faster in 128 x 64 tiles (encoder 6.27 -> 6.00 ms, profiles/r03_gemm_tiles.txt)
  static const int wreg_min = gemm_env("TW_GEMM_WREG_MIN", 512);
  const long long b128 = (long long)((M + 127) / 128) * ((N + 127) / 128);
  int cfg;
  if (ep.mode == EPI_KV_CROSS8) {   // fp8 cross-K/V epilogue: one head per wavefront, i.e. kernel 2 whatever the shape
    if (ElemTraits<T>::kCode != 1 || N % 64 != 0) return hipErrorInvalidValue;
    return gemm_wreg_go<T, 128, 4, 3>(A, amap, W, M, N, K, ep, st);
  }
  // Below that: 128 x 64 tiles on a 3-stage ring, or - up to 1000 rows (one chunk of <= 20 s, two of 10 s) - 64 x 64 tiles on a
  // 4-stage ring (64 KB: two workgroups per CU): one 10 s chunk 3.51 -> 3.04 ms, 15 s 3.90 -> 3.55, 2 x 10 s 4.32 -> 4.24; 30 s and
  // 3 x 10 s lose with it (5.46 -> 5.71, 4.96 -> 5.22) and stay.  Same accumulation order as every other tile shape: results do not
  // change.  Round 4 also tried DEEP rings on the theory that a single stream's K tiles each wait ~2 us for operands nobody has
  // warmed in L2 (8 x 16 KB of a 64 x 64 tile, 6 x 24 KB of a 128 x 64 tile, one workgroup per CU): out-projection 15.5 -> 12.5 us and
  // fc2 39.5 -> 30, but fc1 18.4 -> 34 and QKV 15.3 -> 24.5 - with one workgroup per CU the wide projections run 2.5 rounds and the
  // launch is bound by the aggregate operand traffic of its small tiles (210 MB for fc1 at M = 500 = 6 TB/s), not by latency
  // (profiles/r04_gemm_deep_ring_ab.txt).  TW_GEMM_SMALL_64=0 restores the round-3 choice.
  static const int small64 = gemm_env("TW_GEMM_SMALL_64", 1);
  if (forced >= 0) cfg = forced;
  else if (N % 256 == 0 && b128 >= wreg_min) cfg = (N <= 2048) ? narrow : 5;
  else if (small64 && M <= 1000) cfg = 12;
  else if (M > 64) cfg = 1;
  else cfg = 0;
  if (cfg == 5) {
    // Tile HEIGHT of kernel 2 by how the grid divides over the chip.  Two workgroups share a CU's matrix pipe, so a CU's time is
    // proportional to the rows of the tiles it gets, ceil(tiles / 256) x BM, and the kernel ends with the last CU.  The out-projection
    // and fc2 of 16 x 10 s (M = 8000, N = 1280) are 63 x 5 = 315 tiles of 128 rows: 59 CUs get two and everybody waits for them (the
    // launch runs at the pace of 512 tiles); 100 x 5 tiles of 80 rows put two on (almost) every CU: 160 instead of 256 rows per CU.
    // `ovh` rows stand for the per-tile prologue + epilogue.  TW_GEMM_BM forces a height (80 / 96 / 112 / 128) for A/B runs.
    static const int bm_forced = gemm_env("TW_GEMM_BM", 0);
    static const int ovh = gemm_env("TW_GEMM_BM_OVH", 48);   // 24 picked 96-row tiles for the QKV projection (111 vs 106 us) and lost 3 % at 16 x 30 s
    int best = 128;
    if (bm_forced) {
      best = bm_forced;
    } else {
      long long best_cost = -1;
      for (int bm : {128, 112, 96, 80}) {
        const long long tiles = (long long)((M + bm - 1) / bm) * (N / 256);
        const long long cost = ((tiles + 255) / 256) * (bm + ovh);
        if (best_cost < 0 || cost < best_cost) { best_cost = cost; best = bm; }
      }
    }
    switch (best) {
      case 80: return gemm_wreg_go<T, 80, 4, 3>(A, amap, W, M, N, K, ep, st);
      case 96: return gemm_wreg_go<T, 96, 4, 3>(A, amap, W, M, N, K, ep, st);
      case 112: return gemm_wreg_go<T, 112, 4, 3>(A, amap, W, M, N, K, ep, st);
      default: break;
    }
  }
  switch (cfg) {
    case 5: return gemm_wreg_go<T, 128, 4, 3>(A, amap, W, M, N, K, ep, st);
    case 6: return gemm_wreg_go<T, 128, 4, 2>(A, amap, W, M, N, K, ep, st);
    case 8: return gemm_wreg_go<T, 64, 4, 3>(A, amap, W, M, N, K, ep, st);     // narrow N: twice the workgroups
    case 4: return gemm_go<T, 128, 128, 2, 2, 2>(A, amap, W, M, N, K, ep, st);
    case 1: return gemm_go<T, 128, 64, 2, 2, 3>(A, amap, W, M, N, K, ep, st);
    case 12: return gemm_go<T, 64, 64, 2, 2, 4>(A, amap, W, M, N, K, ep, st);     // 64 KB: two workgroups per CU
    default: return gemm_go<T, 64, 64, 2, 2, 2>(A, amap, W, M, N, K, ep, st);
  }
}

hipError_t launch_gemm(int dtype, const void* A, RowMap amap, const void* W, int M, int N, int K,
                       const GemmEpilogue& ep0, hipStream_t st) {
  static const int staged = gemm_env("TW_GEMM_STAGED", 1);   // 0: the round-3 epilogue everywhere, 2: not for the V^T segment (A/B runs)
  GemmEpilogue ep = ep0;
  ep.staged = staged;
  if (dtype == 1) return gemm_dispatch<bf16_t>(A, amap, W, M, N, K, ep, st);
  if (dtype == 2) return gemm_dispatch<f16_t>(A, amap, W, M, N, K, ep, st);
  return gemm_dispatch<float>(A, amap, W, M, N, K, ep, st);
}
