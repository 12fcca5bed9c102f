// Launch-floor probe: time per kernel of a dependent chain replayed from a hipGraph (MI355X).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k_empty(int* p) {}
__global__ void k_touch(int* p) { if (threadIdx.x == 0 && blockIdx.x == 0) p[0] += 1; }
__global__ void k_stream(const uint4* __restrict__ w, float* __restrict__ o, int n_vec) {
  // every block reads a slice of w once (weight-streaming stand-in), one atomic-free partial per block
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  uint4 acc = {0, 0, 0, 0};
  for (; i < n_vec; i += gridDim.x * blockDim.x) { uint4 v = w[i]; acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w; }
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) o[blockIdx.x] = 1.f;
}
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
int main() {
  int* p; CK(hipMalloc(&p, 4096)); CK(hipMemset(p, 0, 4096));
  const size_t wbytes = 512ull << 20;  // 512 MB so slices come from HBM, not the 256 MB Infinity Cache
  uint4* w; CK(hipMalloc(&w, wbytes)); CK(hipMemset(w, 1, wbytes));
  float* o; CK(hipMalloc(&o, 1 << 20));
  hipStream_t st; CK(hipStreamCreate(&st));
  const int N = 260;
  for (int variant = 0; variant < 6; ++variant) {
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < N; ++i) {
      if (variant == 0) hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, st, p);
      if (variant == 1) hipLaunchKernelGGL(k_empty, dim3(320), dim3(256), 0, st, p);
      if (variant == 2) hipLaunchKernelGGL(k_touch, dim3(320), dim3(256), 0, st, p);
      if (variant >= 3) {
        const size_t mb = variant == 3 ? 3 : variant == 4 ? 13 : 40;  // MB per launch
        const size_t nvec = (mb << 20) / 16;
        const uint4* src = w + ((size_t)i * nvec) % (wbytes / 16 - nvec);
        hipLaunchKernelGGL(k_stream, dim3(variant == 5 ? 1024 : 320), dim3(256), 0, st, src, o, (int)nvec);
      }
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
    const char* names[] = {"empty 1x64", "empty 320x256", "touch 320x256", "stream 3MB 320x256", "stream 13MB 320x256", "stream 40MB 1024x256"};
    printf("%-22s: %.3f us per kernel (graph of %d, %d replays)\n", names[variant], ms * 1e3 / (R * N), N, R);
    // eager
    CK(hipEventRecord(a, st));
    for (int r = 0; r < 5; ++r)
      for (int i = 0; i < N; ++i) hipLaunchKernelGGL(k_touch, dim3(320), dim3(256), 0, st, p);
    CK(hipEventRecord(b, st)); CK(hipStreamSynchronize(st));
    CK(hipEventElapsedTime(&ms, a, b));
    if (variant == 2) printf("%-22s: %.3f us per kernel (eager)\n", "touch 320x256", ms * 1e3 / (5 * N));
    hipGraphExecDestroy(ge); hipGraphDestroy(g);
  }
  return 0;
}
