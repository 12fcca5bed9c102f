// Weight-streaming floor probe (MI355X): what does ONE launch of a decode projection cost when all it does is pull
// its slice of W from HBM with every load in flight at once?  A dependent chain is replayed from a hipGraph (as a decode
// step is); every workgroup stamps s_memrealtime (100 MHz) at entry, when its loads have landed, and at exit, so the
// per-launch span, the dispatch skew and the inter-kernel gap can be separated.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/dbg/stream_floor.hip -o tools/dbg/stream_floor
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int U, bool NT>
__global__ void k_stream(const u32x4* __restrict__ w, float* __restrict__ o, unsigned long long* __restrict__ ts, int launch) {
  const unsigned long long t0 = wall_clock64();
  // block-contiguous slice; wave instruction = 1 KiB contiguous
  const u32x4* p = w + ((size_t)blockIdx.x * U) * blockDim.x + threadIdx.x;
  u32x4 v[U];
#pragma unroll
  for (int u = 0; u < U; ++u) v[u] = NT ? __builtin_nontemporal_load(p + (size_t)u * blockDim.x) : p[(size_t)u * blockDim.x];
  u32x4 acc = v[0];
#pragma unroll
  for (int u = 1; u < U; ++u) acc ^= v[u];
  const unsigned r = acc[0] ^ acc[1] ^ acc[2] ^ acc[3];
  const unsigned long long t1 = wall_clock64() + (r == 0x12345678u ? 1 : 0);
  __syncthreads();
  if (r == 0x12345679u) o[blockIdx.x] = 1.f;
  if (threadIdx.x == 0) {
    unsigned long long* q = ts + ((size_t)launch * gridDim.x + blockIdx.x) * 3;
    q[0] = t0; q[1] = t1; q[2] = wall_clock64();
  }
}

struct Cfg { const char* name; int blocks, threads, U; bool nt; };

template <int U, bool NT>
void launch(const Cfg& c, const u32x4* src, float* o, unsigned long long* ts, int i, hipStream_t st) {
  hipLaunchKernelGGL((k_stream<U, NT>), dim3(c.blocks), dim3(c.threads), 0, st, src, o, ts, i);
}

int main() {
  const size_t wbytes = 1024ull << 20;
  u32x4* w; CK(hipMalloc(&w, wbytes)); CK(hipMemset(w, 1, wbytes));
  float* o; CK(hipMalloc(&o, 1 << 20));
  const int N = 64;
  unsigned long long* ts; CK(hipMalloc(&ts, (size_t)N * 2048 * 3 * 8));
  hipStream_t st; CK(hipStreamCreate(&st));
  const Cfg cfgs[] = {
      {"3.3MB  80x512 U5", 80, 512, 5, false},    {"3.3MB  80x512 U5 nt", 80, 512, 5, true},
      {"3.3MB 160x256 U5", 160, 256, 5, false},   {"3.3MB 400x256 U2 nt", 400, 256, 2, true},
      {"3.3MB 200x1024 U1 nt", 200, 1024, 1, true},
      {"9.8MB 240x512 U5", 240, 512, 5, false},   {"9.8MB 240x512 U5 nt", 240, 512, 5, true},
      {"9.8MB 600x512 U2 nt", 600, 512, 2, true},
      {"13MB  320x512 U5", 320, 512, 5, false},   {"13MB  320x512 U5 nt", 320, 512, 5, true},
      {"13MB  640x256 U5 nt", 640, 256, 5, true}, {"13MB  160x512 U10 nt", 160, 512, 10, true},
      {"13MB  256x512 U6 nt", 256, 512, 6, true}, {"13MB  800x512 U2 nt", 800, 512, 2, true},
      {"13MB  400x1024 U2 nt", 400, 1024, 2, true}, {"13MB  1600x256 U2 nt", 1600, 256, 2, true},
      {"26MB  640x512 U5 nt", 640, 512, 5, true}, {"26MB  256x1024 U6 nt", 256, 1024, 6, true},
      {"42MB 1024x512 U5 nt", 1024, 512, 5, true},
  };
  for (const Cfg& c : cfgs) {
    const size_t nvec = (size_t)c.blocks * c.threads * c.U;
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < N; ++i) {
      const u32x4* src = w + ((size_t)i * nvec) % (wbytes / 16 - nvec);
#define L(UU) if (c.U == UU) { if (c.nt) launch<UU, true>(c, src, o, ts, i, st); else launch<UU, false>(c, src, o, ts, i, st); }
      L(1) L(2) L(5) L(6) L(10)
#undef L
    }
    CK(hipStreamEndCapture(st, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    CK(hipGraphLaunch(ge, st)); CK(hipStreamSynchronize(st));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    CK(hipEventRecord(a, st));
    const int R = 20;
    for (int r = 0; r < R; ++r) CK(hipGraphLaunch(ge, st));
    CK(hipEventRecord(b, st)); CK(hipStreamSynchronize(st));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    std::vector<unsigned long long> h((size_t)N * c.blocks * 3);
    CK(hipMemcpy(h.data(), ts, h.size() * 8, hipMemcpyDeviceToHost));
    double span = 0, skew = 0, gap = 0, lat = 0; unsigned long long prev_end = 0;
    for (int i = 0; i < N; ++i) {
      unsigned long long s0 = ~0ull, s1 = 0, e1 = 0; double l = 0;
      for (int bk = 0; bk < c.blocks; ++bk) {
        const unsigned long long* q = &h[((size_t)i * c.blocks + bk) * 3];
        s0 = std::min(s0, q[0]); s1 = std::max(s1, q[0]); e1 = std::max(e1, q[2]); l += (double)(q[1] - q[0]);
      }
      span += (double)(e1 - s0); skew += (double)(s1 - s0); lat += l / c.blocks;
      if (i > 0) gap += (double)((long long)(s0 - prev_end));
      prev_end = e1;
    }
    const double us = ms * 1e3 / (R * N), mb = nvec * 16 / 1e6;
    printf("%-22s: %6.2f us/launch (%5.2f TB/s) | span %.2f us, dispatch skew %.2f, mean load-wait %.2f, gap %.2f\n", c.name, us,
           mb / us, span / N * 0.01, skew / N * 0.01, lat / N * 0.01, gap / (N - 1) * 0.01);
    hipGraphExecDestroy(ge); hipGraphDestroy(g);
  }
  return 0;
}
