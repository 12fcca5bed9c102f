// A3: encoder self-attention (non-causal, no mask, head_dim 64, T <= 1500) as a flash-style MFMA kernel.
//
// Inputs come straight from the fused QKV GEMM epilogue: q (already scaled by 1/8, the scale is
// folded into q_proj at load) and k as [B,H,T,64], v TRANSPOSED as vt[B,H,64,Tp] (Tp = T rounded up
// to 64, pad region zero) so that the P.V contraction over keys is K-contiguous for both MFMA
// operands - no in-kernel transpose, no tr-reads.
//
// Orientation ("swapped" QK^T, cf. cdna_hip_programming.md T12): S^T = K.Q^T with the K tile as the
// MFMA A operand and the wave's Q fragment as B, so a lane owns ONE query (lane&15) and 4 keys per
// 16-key sub-tile.  Row max / row sum are then in-lane reductions plus two wavefront shuffles
// (xor 16, xor 32), and the probabilities feed the P.V MFMA (O^T = Vt.P^T) directly from registers:
// the contraction index of an MFMA is order-free, so each lane's 8 keys {16a+4kb..+3, 16(a+1)+4kb..+3}
// form its k-slice and the matching Vt fragment is two 8-byte LDS reads.
//
// LDS: K tile as [16-B slot][key ^ slot] (conflict-free ds_write_b128 / ds_read_b128, same scheme as
// the GEMM), Vt tile as [d][64 keys] rows padded by 16 B (conflict-free ds_read_b64).  Both tiles are
// register-prefetched one tile ahead and double-buffered in LDS: one barrier per 64-key tile.
#include "tw_common.h"

#include <cstdlib>

namespace {

template <typename T> struct AttnTraits;
template <> struct AttnTraits<bf16_t> {
  static constexpr int E = 8;        // elements per 16 B
  static constexpr int NSLOT = 8;    // 16-B slots per 64-dim row
  static constexpr int KK = 2;       // MFMA k-steps over head_dim
};
template <> struct AttnTraits<f16_t> : AttnTraits<bf16_t> {};
template <> struct AttnTraits<float> {
  static constexpr int E = 4;
  static constexpr int NSLOT = 16;
  static constexpr int KK = 4;
};

template <typename T>
__device__ __forceinline__ f32x4_t mfma16(const u32x4_t& a, const u32x4_t& b, f32x4_t acc) {
  return tw_mfma32<T>(a, b, acc);   // bf16 / f16
}
template <>
__device__ __forceinline__ f32x4_t mfma16<float>(const u32x4_t& a, const u32x4_t& b, f32x4_t acc) {
  const f32x4_t af = __builtin_bit_cast(f32x4_t, a);
  const f32x4_t bf = __builtin_bit_cast(f32x4_t, b);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(af[0], bf[0], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(af[1], bf[1], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(af[2], bf[2], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(af[3], bf[3], acc, 0, 0, 0);
  return acc;
}

// exp(x) for x <= 0 in the softmax.  Strict-f32 contexts keep the library expf (parity to 1e-6); bf16 contexts use the hardware
// 2^x on x*log2(e) - one multiply and one transcendental instead of the library's range handling (the probabilities are rounded to
// 8 mantissa bits for the P.V operand anyway).  The kernel is VALU-bound: per 64-key tile a wavefront issues ~34 exponentials next
// to 32 MFMAs.
template <typename T>
__device__ __forceinline__ float attn_exp(float x) {
  if constexpr (sizeof(T) == 4) return expf(x);
  else return __builtin_amdgcn_exp2f(x * 1.4426950408889634f);
}


// Workgroup id -> (stream*head, query block).  All query blocks of one head read the same K / V^T (128 KB at T = 500); consecutive
// workgroup ids go to the 8 XCDs round-robin (MI355X_MICROARCH.md), so with the query block as the fastest grid index every
// head's K / V^T was fetched into 4 different L2s.  xcd = 1: head bh runs on XCD bh % 8, its query blocks back to back there.
template <typename T, int QT>
__global__ __launch_bounds__(256, 2) void enc_attn_kernel(const T* __restrict__ q, const T* __restrict__ k,
                                                        const T* __restrict__ vt, T* __restrict__ out, int H, int Tlen,
                                                        int Tp, int nbh, int nq, int xcd) {
  using TR = AttnTraits<T>;
  constexpr int E = TR::E, NSLOT = TR::NSLOT, KK = TR::KK;
  constexpr int KV = 64 * NSLOT / 256;          // K-tile vectors per thread
  constexpr int VROW = 64 * (int)sizeof(T) + 16;  // padded Vt row bytes
  constexpr int VVEC = 64 * (int)sizeof(T) / 16;  // 16-B vectors per Vt row (= NSLOT)
  constexpr int VV = 64 * VVEC / 256;             // Vt-tile vectors per thread
  __shared__ u32x4_t ks[2][NSLOT * 64];
  __shared__ __attribute__((aligned(16))) unsigned char vs[2][64 * VROW];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 15, kb = lane >> 4;
  int bhi, qb;
  if (xcd) {
    const int local = blockIdx.x >> 3;
    bhi = (local / nq) * 8 + (blockIdx.x & 7);
    qb = local % nq;
  } else {
    bhi = blockIdx.x / nq;
    qb = blockIdx.x % nq;
  }
  if (bhi >= nbh) return;   // padding ids of the last group of 8 heads (whole workgroup, before any barrier)
  const int b = bhi / H, h = bhi - b * H;
  const int qblock = qb * (64 * QT);
  const long long bh = bhi;
  const T* qh = q + bh * Tlen * 64;
  const T* kh = k + bh * Tlen * 64;
  const T* vh = vt + bh * 64 * Tp;

  // Q fragments (B operand: j = query = lane&15, k-slot = kk*4 + kb)
  u32x4_t qf[QT][KK];
#pragma unroll
  for (int t = 0; t < QT; ++t) {
    int qi = qblock + (wave * QT + t) * 16 + fr;
    if (qi >= Tlen) qi = Tlen - 1;
#pragma unroll
    for (int kk = 0; kk < KK; ++kk)
      qf[t][kk] = *reinterpret_cast<const u32x4_t*>(qh + (long long)qi * 64 + (kk * 4 + kb) * E);
  }

  f32x4_t o[QT][4];
  float mrun[QT], lrun[QT];
#pragma unroll
  for (int t = 0; t < QT; ++t) {
    mrun[t] = -1.0e30f;
    lrun[t] = 0.f;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) o[t][dt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  }

  const int ntiles = (Tlen + 63) / 64;
  u32x4_t kreg[KV], vreg[VV];
  auto gload = [&](int tile) {
    const int key0 = tile * 64;
#pragma unroll
    for (int i = 0; i < KV; ++i) {
      const int v = i * 256 + tid;
      int key = key0 + v / NSLOT;
      const int slot = v % NSLOT;
      if (key >= Tlen) key = Tlen - 1;
      kreg[i] = *reinterpret_cast<const u32x4_t*>(kh + (long long)key * 64 + slot * E);
    }
#pragma unroll
    for (int i = 0; i < VV; ++i) {
      const int v = i * 256 + tid;
      const int d = v / VVEC, j = v % VVEC;
      vreg[i] = *reinterpret_cast<const u32x4_t*>(vh + (long long)d * Tp + key0 + j * E);
    }
  };
  auto lstore = [&](int buf) {
#pragma unroll
    for (int i = 0; i < KV; ++i) {
      const int v = i * 256 + tid;
      const int key = v / NSLOT, slot = v % NSLOT;
      ks[buf][slot * 64 + (key ^ slot)] = kreg[i];
    }
#pragma unroll
    for (int i = 0; i < VV; ++i) {
      const int v = i * 256 + tid;
      const int d = v / VVEC, j = v % VVEC;
      *reinterpret_cast<u32x4_t*>(&vs[buf][d * VROW + j * 16]) = vreg[i];
    }
  };

  gload(0);
  lstore(0);
  __syncthreads();
  int buf = 0;
  for (int tile = 0; tile < ntiles; ++tile) {
    const bool more = tile + 1 < ntiles;
    if (more) gload(tile + 1);
    const int key0 = tile * 64;

    // ---- S^T = K.Q^T ----
    f32x4_t s[QT][4];
#pragma unroll
    for (int t = 0; t < QT; ++t)
#pragma unroll
      for (int kt = 0; kt < 4; ++kt) s[t][kt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kk = 0; kk < KK; ++kk) {
      const int slot = kk * 4 + kb;
#pragma unroll
      for (int kt = 0; kt < 4; ++kt) {
        const u32x4_t kf = ks[buf][slot * 64 + ((kt * 16 + fr) ^ slot)];
#pragma unroll
        for (int t = 0; t < QT; ++t) s[t][kt] = mfma16<T>(kf, qf[t][kk], s[t][kt]);
      }
    }

    // ---- online softmax (per lane: one query, keys key0 + 16kt + 4kb + r) ----
    // The loop is VALU-bound (round-4 counters: 384 VALU instructions per 64-key tile and wavefront next to 32 MFMAs, 43 % of the
    // wave cycles issuing): keys are masked only in the one tile that has padding (wave-uniform branch), and the per-score
    // arithmetic is written on float pairs so that it compiles to v_pk_fma_f32 / v_pk_add_f32 / v_pk_mul_f32 (two scores per
    // instruction) and v_max3_f32.
    const bool partial = key0 + 64 > Tlen;
    float alpha[QT];
#pragma unroll
    for (int t = 0; t < QT; ++t) {
      if (partial) {
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (key0 + kt * 16 + kb * 4 + r >= Tlen) s[t][kt][r] = -1.0e30f;
      }
      float mx = -1.0e30f;
#pragma unroll
      for (int kt = 0; kt < 4; ++kt) {
        mx = fmaxf(fmaxf(mx, s[t][kt][0]), s[t][kt][1]);
        mx = fmaxf(fmaxf(mx, s[t][kt][2]), s[t][kt][3]);
      }
      mx = tw_xor32_max(tw_xor16_max(mx));
      const float mnew = fmaxf(mrun[t], mx);
      alpha[t] = attn_exp<T>(mrun[t] - mnew);
      mrun[t] = mnew;
      float ps = 0.f;
      if constexpr (sizeof(T) == 4) {
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float p = expf(s[t][kt][r] - mnew);
            s[t][kt][r] = p;
            ps += p;
          }
      } else {
        // exp(s - m) = 2^(s * log2(e) - m * log2(e)): one fused multiply-add per PAIR of scores and the hardware 2^x per score
        const float L2E = 1.4426950408889634f;
        const f32x2_t c2 = {L2E, L2E};
        const f32x2_t nml = {-mnew * L2E, -mnew * L2E};
        f32x2_t ps2 = {0.f, 0.f};
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
          for (int h2 = 0; h2 < 2; ++h2) {
            f32x2_t v = {s[t][kt][2 * h2], s[t][kt][2 * h2 + 1]};
            v = v * c2 + nml;
            f32x2_t p = {__builtin_amdgcn_exp2f(v[0]), __builtin_amdgcn_exp2f(v[1])};
            s[t][kt][2 * h2] = p[0];
            s[t][kt][2 * h2 + 1] = p[1];
            ps2 += p;
          }
        ps = ps2[0] + ps2[1];
      }
      lrun[t] = lrun[t] * alpha[t] + ps;
      const f32x4_t a4 = {alpha[t], alpha[t], alpha[t], alpha[t]};
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) o[t][dt] *= a4;
    }

    // ---- O^T += Vt.P^T ----
    if constexpr (sizeof(T) == 2) {
#pragma unroll
      for (int pr = 0; pr < 2; ++pr) {  // key sub-tile pairs (0,1), (2,3)
        u32x4_t pf[QT];
#pragma unroll
        for (int t = 0; t < QT; ++t) {
          pf[t][0] = tw_pack2_inrange<T>(s[t][2 * pr][0], s[t][2 * pr][1]);
          pf[t][1] = tw_pack2_inrange<T>(s[t][2 * pr][2], s[t][2 * pr][3]);
          pf[t][2] = tw_pack2_inrange<T>(s[t][2 * pr + 1][0], s[t][2 * pr + 1][1]);
          pf[t][3] = tw_pack2_inrange<T>(s[t][2 * pr + 1][2], s[t][2 * pr + 1][3]);
        }
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
          const unsigned char* row = &vs[buf][(dt * 16 + fr) * VROW];
          const u32x2_t v0 = *reinterpret_cast<const u32x2_t*>(row + ((2 * pr) * 16 + kb * 4) * 2);
          const u32x2_t v1 = *reinterpret_cast<const u32x2_t*>(row + ((2 * pr + 1) * 16 + kb * 4) * 2);
          const u32x4_t vf = u32x4_t{v0[0], v0[1], v1[0], v1[1]};
#pragma unroll
          for (int t = 0; t < QT; ++t) o[t][dt] = mfma16<T>(vf, pf[t], o[t][dt]);
        }
      }
    } else {
#pragma unroll
      for (int kt = 0; kt < 4; ++kt) {
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
          const unsigned char* row = &vs[buf][(dt * 16 + fr) * VROW];
          const u32x4_t vf = *reinterpret_cast<const u32x4_t*>(row + (kt * 16 + kb * 4) * 4);
#pragma unroll
          for (int t = 0; t < QT; ++t) o[t][dt] = mfma16<T>(vf, __builtin_bit_cast(u32x4_t, s[t][kt]), o[t][dt]);
        }
      }
    }

    if (more) lstore(buf ^ 1);
    __syncthreads();
    buf ^= 1;
  }

  // ---- normalise and store: lane holds d = 16dt + 4kb + r of query fr ----
  const int dmodel = H * 64;
#pragma unroll
  for (int t = 0; t < QT; ++t) {
    float l = lrun[t];
    l = tw_xor32_sum(tw_xor16_sum(l));
    const float inv = 1.0f / l;
    const int qi = qblock + (wave * QT + t) * 16 + fr;
    if (qi >= Tlen) continue;
    T* orow = out + ((long long)b * Tlen + qi) * dmodel + h * 64;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      T* p = orow + dt * 16 + kb * 4;
      if constexpr (sizeof(T) == 2) {
        u32x2_t w;
        w[0] = tw_pack2_inrange<T>(o[t][dt][0] * inv, o[t][dt][1] * inv);
        w[1] = tw_pack2_inrange<T>(o[t][dt][2] * inv, o[t][dt][3] * inv);
        *reinterpret_cast<u32x2_t*>(p) = w;
      } else {
        *reinterpret_cast<f32x4_t*>(p) = f32x4_t{o[t][dt][0] * inv, o[t][dt][1] * inv, o[t][dt][2] * inv, o[t][dt][3] * inv};
      }
    }
  }
}

}  // namespace

hipError_t launch_enc_attention(int dtype, const void* q, const void* k, const void* vt, void* out, int B, int H,
                                int T, int Tp, hipStream_t st) {
  if (B <= 0 || T <= 0 || Tp % 64 != 0 || Tp < T) return hipErrorInvalidValue;
  // 128 queries per workgroup when that still yields >= 2 workgroups per CU, else 64.
  const long long blocks128 = (long long)B * H * ((T + 127) / 128);
  static const int qb_env = []() { const char* e = getenv("TW_ATTN_QB"); return e ? atoi(e) : 0; }();   // diagnostics: 1 / 2 forces 64 / 128 queries
  const bool big = qb_env ? qb_env == 2 : blocks128 >= 512;
  // TW_ATTN_XCD: 1 (default) = heads pinned to XCDs when there are at least 64 of them (with the 20 heads of one stream 4 XCDs
  // would get three heads and 4 two: 5.68 -> 5.88 ms for one 30 s chunk), 0 = never, 2 = always
  static const int xcd_env = []() { const char* e = getenv("TW_ATTN_XCD"); return e ? atoi(e) : 1; }();
  const int nbh = B * H;
  const int xcd = (xcd_env == 2 || (xcd_env == 1 && nbh >= 64)) ? 1 : 0;
  auto grid_for = [&](int nq) { return dim3((unsigned)((xcd ? 8 * ((nbh + 7) / 8) : nbh) * nq)); };
  if (dtype == 1) {
    if (big)
      hipLaunchKernelGGL((enc_attn_kernel<bf16_t, 2>), grid_for((T + 127) / 128), dim3(256), 0, st, (const bf16_t*)q,
                         (const bf16_t*)k, (const bf16_t*)vt, (bf16_t*)out, H, T, Tp, nbh, (T + 127) / 128, xcd);
    else
      hipLaunchKernelGGL((enc_attn_kernel<bf16_t, 1>), grid_for((T + 63) / 64), dim3(256), 0, st, (const bf16_t*)q,
                         (const bf16_t*)k, (const bf16_t*)vt, (bf16_t*)out, H, T, Tp, nbh, (T + 63) / 64, xcd);
  } else if (dtype == 2) {
    if (big)
      hipLaunchKernelGGL((enc_attn_kernel<f16_t, 2>), grid_for((T + 127) / 128), dim3(256), 0, st, (const f16_t*)q,
                         (const f16_t*)k, (const f16_t*)vt, (f16_t*)out, H, T, Tp, nbh, (T + 127) / 128, xcd);
    else
      hipLaunchKernelGGL((enc_attn_kernel<f16_t, 1>), grid_for((T + 63) / 64), dim3(256), 0, st, (const f16_t*)q,
                         (const f16_t*)k, (const f16_t*)vt, (f16_t*)out, H, T, Tp, nbh, (T + 63) / 64, xcd);
  } else {
    hipLaunchKernelGGL((enc_attn_kernel<float, 1>), grid_for((T + 63) / 64), dim3(256), 0, st, (const float*)q,
                       (const float*)k, (const float*)vt, (float*)out, H, T, Tp, nbh, (T + 63) / 64, xcd);
  }
  return hipGetLastError();
}
