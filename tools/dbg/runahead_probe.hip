// Run-ahead probe (MI355X): what does a dependent chain of decode-projection-like launches cost when launch k+1 is already
// RESIDENT (its weight slice in flight / in registers) while launch k computes, instead of starting at a kernel boundary?
//
//   mode 0  baseline: one stream, a kernel boundary between consecutive ops (what the decode step does today)
//   mode 1  run-ahead: ops alternate between NS streams (hardware queues); op k+1 requests its weights, then polls the
//           per-workgroup flags op k publishes after its write-through stores; the activation block is read with sc1 loads
//
// Every op: G workgroups x 512 threads; a workgroup streams U x 8 KiB of "weights" (non-temporal), waits for its
// dependency, reads the 40 KB activation block (as 5 x 1 KiB per wavefront, like skinny_mfma_kernel), reduces through LDS
// (one barrier) and wave 0 writes the workgroup's 512-byte slice of the next activation block.  Results are checked: every
// op adds 1 to every activation element, so a stale read anywhere shows in the final block.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/dbg/runahead_probe.hip -o tools/dbg/runahead_probe
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

constexpr int ACT_WORDS = 10240;   // 40 KB activation block (16 streams x 1280 bf16)
constexpr int SPIN_MAX = 20000;   // bounded spin: give up (and flag the error) rather than hang the box

struct OpArgs {
  const u32x4* w;            // this op's weight slice base
  const unsigned* x;         // activation block in
  unsigned* y;               // activation block out
  const unsigned* dep_flags; // flags of the op this one depends on (null = none)
  unsigned* my_flags;        // one word per workgroup of this op
  int dep_g;                 // workgroups of the dependency
  unsigned epoch;
  unsigned* err;
  unsigned long long* ts;    // [G][4] stamps
  const u32x4* w_next;       // mode 5/6: the NEXT op's weight slice, pulled towards the chip (Infinity Cache) by this op
  int next_u;                // ... its fragments per thread
  int nt;                    // own weight loads non-temporal (1) or default policy (0)
};

template <int U, bool RA>
__global__ __launch_bounds__(512) void k_op(OpArgs a) {
  __shared__ unsigned part[8][64];
  __shared__ unsigned first[8][64];
  __shared__ unsigned go;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const unsigned long long t0 = wall_clock64();
  // ---- weights: everything in flight at once ----
  const u32x4* p = a.w + ((size_t)blockIdx.x * U) * 512 + tid;
  u32x4 v[U];
  if (a.nt) {
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = __builtin_nontemporal_load(p + (size_t)u * 512);
  } else {
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = p[(size_t)u * 512];
  }
  __builtin_amdgcn_sched_barrier(0);
  // prefetch of the next op's slice (same block index, same shape): requested behind this op's own loads with the default cache
  // policy, consumed (discarded) at the very end.  What is measured: does the next launch start faster when its weights are already
  // on the chip (Infinity Cache; the per-XCD L2 is invalidated at the kernel boundary)?
  constexpr int PF = U <= 5 ? U : 1;
  u32x4 pf[PF] = {};
  if (a.w_next) {
    const u32x4* q = a.w_next + ((size_t)blockIdx.x * U) * 512 + tid;
#pragma unroll
    for (int u = 0; u < PF; ++u) pf[u] = q[(size_t)u * 512];
  }
  __builtin_amdgcn_sched_barrier(0);
  // ---- dependency ----
  if (RA) {
    if (a.dep_flags) {
      if (wave == 0) {
        bool ok = false;
        for (int spin = 0; spin < SPIN_MAX && !ok; ++spin) {
          bool mine = true;
          for (int i = lane; i < a.dep_g; i += 64)
            mine &= __hip_atomic_load(a.dep_flags + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == a.epoch - 1u;   // the producer's epoch
          ok = __all(mine);
          if (!ok) __builtin_amdgcn_s_sleep(2);
        }
        if (!ok && lane == 0) atomicAdd(a.err, 1u);
      }
      __syncthreads();
    }
  }
  const unsigned long long t1 = wall_clock64();
  // ---- activations: 5 x 1 KiB per wavefront (the block is 40 KB = 8 wavefronts x 5 KiB) ----
  const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned*>(a.x), 0, ACT_WORDS * 4, 0x00020000);
  u32x4 xv[5];
#pragma unroll
  for (int i = 0; i < 5; ++i)
    xv[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(xr, (unsigned)(((wave * 5 + i) * 64 + lane) * 16), 0, RA ? 16 : 0));
  unsigned acc = 0;   // words of this lane that differ from the block's (uniform) value as ITS first word shows it
#pragma unroll
  for (int i = 0; i < 5; ++i)
#pragma unroll
    for (int q = 0; q < 4; ++q) acc += xv[i][q] != xv[0][0] ? 1u : 0u;
  if (tid == 0) go = xv[0][0];
  unsigned wsum = 0;
#pragma unroll
  for (int u = 0; u < U; ++u) wsum ^= v[u][0] ^ v[u][1] ^ v[u][2] ^ v[u][3];
  part[wave][lane] = acc + (wsum == 0x12345678u ? 1u : 0u);
  first[wave][lane] = xv[0][0];
  const unsigned long long t2 = wall_clock64();
  __syncthreads();
  // ---- epilogue: wave 0 writes this workgroup's slice (ACT_WORDS / G words) of the next block: value = x[idx] + 1 ----
  if (wave == 0) {
    unsigned s = 0;
#pragma unroll
    for (int w = 0; w < 8; ++w) s += part[w][lane] + (first[w][lane] != go ? 1u : 0u);
    const int per = ACT_WORDS / gridDim.x;          // words per workgroup (G divides ACT_WORDS)
    const int base = blockIdx.x * per;
    // every lane recomputes "its" outputs from the block it read: out = in + 1 (in is uniform, so in = s-derived check value)
    const unsigned in0 = go;                         // all words of the block are equal by construction
    if (__any(s != 0u || xv[0][0] != in0) && lane == 0) atomicAdd(a.err + 1, 1u);   // some word of the block was stale
    const unsigned outv = in0 + 1u + (s == 0xdeadbeefu ? 1u : 0u);
    for (int i = lane * 2; i < per; i += 128) {
      u32x2 o2 = {outv, outv};
      if (RA) __hip_atomic_store(reinterpret_cast<unsigned long long*>(a.y + base + i), __builtin_bit_cast(unsigned long long, o2),
                                 __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      else *reinterpret_cast<u32x2*>(a.y + base + i) = o2;
    }
    if (RA) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (lane == 0) __hip_atomic_store(a.my_flags + blockIdx.x, a.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  if (a.w_next) {
    unsigned z = 0;
#pragma unroll
    for (int u = 0; u < PF; ++u) z ^= pf[u][0] ^ pf[u][3];
    if (z == 0x0badf00du) a.y[0] = z;   // keeps the prefetch loads alive to the end of the kernel
  }
  if (tid == 0 && a.ts) {
    unsigned long long* q = a.ts + (size_t)blockIdx.x * 4;
    q[0] = t0; q[1] = t1; q[2] = t2; q[3] = wall_clock64();
  }
}

struct Cfg { const char* name; int G, U; };

template <int U>
static void launch_op(bool ra, int G, const OpArgs& a, hipStream_t st) {
  if (ra) hipLaunchKernelGGL((k_op<U, true>), dim3(G), dim3(512), 0, st, a);
  else hipLaunchKernelGGL((k_op<U, false>), dim3(G), dim3(512), 0, st, a);
}
static void launch_any(int U, bool ra, int G, const OpArgs& a, hipStream_t st) {
  if (U == 2) launch_op<2>(ra, G, a, st);
  else if (U == 5) launch_op<5>(ra, G, a, st);
  else if (U == 16) launch_op<16>(ra, G, a, st);
}

int main(int argc, char** argv) {
  const size_t wbytes = 2048ull << 20;
  u32x4* w; CK(hipMalloc(&w, wbytes)); CK(hipMemset(w, 1, wbytes));
  const int N = 56;                       // ops per replay (8 decoder layers x 7)
  unsigned *act, *flags, *err; unsigned long long* ts;
  CK(hipMalloc(&act, (size_t)(N + 1) * ACT_WORDS * 4));
  CK(hipMalloc(&flags, (size_t)N * 512 * 4));
  CK(hipMalloc(&err, 8)); CK(hipMemset(err, 0, 8));
  CK(hipMalloc(&ts, (size_t)N * 512 * 4 * 8));
  hipStream_t sts[4];
  for (int i = 0; i < 4; ++i) CK(hipStreamCreateWithFlags(&sts[i], hipStreamNonBlocking));
  hipEvent_t fork_ev, join_ev[4], a, b;
  CK(hipEventCreateWithFlags(&fork_ev, hipEventDisableTiming));
  for (int i = 0; i < 4; ++i) CK(hipEventCreateWithFlags(&join_ev[i], hipEventDisableTiming));
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));

  // layer-like mix: QKV 320x5, self-attn (tiny) 320x2... kept simple: uniform chains + one mixed chain
  const Cfg cfgs[] = {{"13MB 320x512 U5", 320, 5}, {"6.6MB 160x512 U5", 160, 5}, {"3.3MB 160x512 U2(+)", 160, 2},
                      {"42MB 320x512 U16", 320, 16}};
  const int mixG[7] = {320, 320, 160, 320, 160, 320, 160};
  const int mixU[7] = {5, 2, 5, 16, 2, 5, 5};   // QKV, self-attn, out-proj, cross-attn, cross-out, fc1, fc2(as 160 x 80 KB ~ U5 x2: kept U5)
  const int n_cfg = (int)(sizeof(cfgs) / sizeof(cfgs[0])) + 1;

  for (int ci = 0; ci < n_cfg; ++ci) {
    const bool mixed = ci == n_cfg - 1;
    for (int mode = 0; mode < 8; ++mode) {
      if (const char* sel = getenv("RP_MODES")) { char key[3] = {(char)('0' + mode), 0, 0}; if (!strstr(sel, key)) continue; }
      if (mode >= 5 && (mixed || cfgs[ci].U > 5)) continue;   // prefetch modes: uniform chains of the projection-like shapes
      // mode 0: boundary chain in a graph; 1: run-ahead 2 streams in a graph; 2: run-ahead 3 streams in a graph;
      // 3: run-ahead 2 streams, direct launches (no graph); 4: run-ahead kernels on ONE stream (flags + boundaries: overhead of the protocol)
      // 5: boundary chain, every op prefetches the next op's slice, own loads default policy; 6: same, own loads nt; 7: boundary chain, default policy, no prefetch
      const bool ra = mode >= 1 && mode <= 4;
      const int NS = (mode == 0 || mode >= 4) ? 1 : (mode == 2 ? 3 : 2);
      const bool graph = mode != 3;
      auto enqueue = [&](bool with_ts) -> int {
        CK(hipMemsetAsync(flags, 0, (size_t)N * 512 * 4, sts[0]));
        CK(hipMemsetAsync(act, 0, ACT_WORDS * 4, sts[0]));
        if (NS > 1) {
          CK(hipEventRecord(fork_ev, sts[0]));
          for (int s = 1; s < NS; ++s) CK(hipStreamWaitEvent(sts[s], fork_ev, 0));
        }
        size_t woff = 0;
        int prevG = 0;
        for (int i = 0; i < N; ++i) {
          const int G = mixed ? mixG[i % 7] : cfgs[ci].G, U = mixed ? mixU[i % 7] : cfgs[ci].U;
          OpArgs o{};
          o.w = w + woff; woff += (size_t)G * U * 512; if (woff + 320 * 16 * 512 > wbytes / 16) woff = 0;
          o.x = act + (size_t)i * ACT_WORDS; o.y = act + (size_t)(i + 1) * ACT_WORDS;
          o.dep_flags = i > 0 ? flags + (size_t)(i - 1) * 512 : nullptr; o.dep_g = prevG;
          o.my_flags = flags + (size_t)i * 512; o.epoch = (unsigned)(i + 1); o.err = err;
          o.ts = with_ts ? ts + (size_t)i * 512 * 4 : nullptr;
          o.nt = (mode == 5 || mode == 7) ? 0 : 1;
          o.next_u = U;
          o.w_next = ((mode == 5 || mode == 6) && i + 1 < N && woff + (size_t)G * U * 512 * 2 < wbytes / 16) ? w + woff : nullptr;   // woff already points at the next op's slice
          launch_any(U, ra, G, o, sts[i % NS]);
          prevG = G;
        }
        if (NS > 1) {
          for (int s = 1; s < NS; ++s) { CK(hipEventRecord(join_ev[s], sts[s])); CK(hipStreamWaitEvent(sts[0], join_ev[s], 0)); }
        }
        return 0;
      };
      hipGraph_t g = nullptr; hipGraphExec_t ge = nullptr;
      if (graph) {
        CK(hipStreamBeginCapture(sts[0], hipStreamCaptureModeThreadLocal));
        if (enqueue(true)) return 1;
        CK(hipStreamEndCapture(sts[0], &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
      }
      auto run = [&]() -> int { if (graph) { CK(hipGraphLaunch(ge, sts[0])); } else { if (enqueue(true)) return 1; } return 0; };
      for (int r = 0; r < 3; ++r) if (run()) return 1;
      CK(hipStreamSynchronize(sts[0]));
      CK(hipEventRecord(a, sts[0]));
      const int R = 20;
      for (int r = 0; r < R; ++r) if (run()) return 1;
      CK(hipEventRecord(b, sts[0])); CK(hipStreamSynchronize(sts[0]));
      float ms; CK(hipEventElapsedTime(&ms, a, b));
      // correctness: the last block must hold N everywhere
      std::vector<unsigned> h(ACT_WORDS); unsigned herr[2] = {0, 0};
      CK(hipMemcpy(h.data(), act + (size_t)N * ACT_WORDS, ACT_WORDS * 4, hipMemcpyDeviceToHost));
      CK(hipMemcpy(herr, err, 8, hipMemcpyDeviceToHost));
      int bad = 0;
      for (int i = 0; i < ACT_WORDS; ++i) bad += h[i] != (unsigned)N;
      // stamps: mean over ops of (first wg start -> last wg end), wait time, and start-to-start distance
      std::vector<unsigned long long> hs((size_t)N * 512 * 4);
      CK(hipMemcpy(hs.data(), ts, hs.size() * 8, hipMemcpyDeviceToHost));
      double span = 0, wait = 0, post = 0, s2s = 0; unsigned long long prev_end = 0; double e2e = 0;
      for (int i = 0; i < N; ++i) {
        const int G = mixed ? mixG[i % 7] : cfgs[ci].G;
        unsigned long long s0 = ~0ull, e1 = 0; double wsum = 0, psum = 0;
        for (int bk = 0; bk < G; ++bk) {
          const unsigned long long* q = &hs[((size_t)i * 512 + bk) * 4];
          s0 = std::min(s0, q[0]); e1 = std::max(e1, q[3]); wsum += (double)(q[1] - q[0]); psum += (double)(q[3] - q[1]);
        }
        span += (double)(e1 - s0); wait += wsum / G; post += psum / G;
        if (i > 0) e2e += (double)((long long)(e1 - prev_end));
        prev_end = e1;
      }
      (void)s2s;
      const char* mname[] = {"boundary/graph", "runahead2/graph", "runahead3/graph", "runahead2/direct", "flags+boundary/graph",
                             "prefetch-next/graph", "prefetch-next nt/graph", "boundary default-policy"};
      printf("%-22s %-22s: %6.2f us/op | resident span %.2f, entry->go %.2f, go->exit %.2f, end-to-end step %.2f | bad %d timeouts %u stale %u\n",
             mixed ? "mixed layer (7 ops)" : cfgs[ci].name, mname[mode], ms * 1e3 / (R * N), span / N * 0.01, wait / N * 0.01,
             post / N * 0.01, e2e / (N - 1) * 0.01, bad, herr[0], herr[1]);
      fflush(stdout);
      CK(hipMemset(err, 0, 8));
      if (graph) { hipGraphExecDestroy(ge); hipGraphDestroy(g); }
    }
  }
  return 0;
}
