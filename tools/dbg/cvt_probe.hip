#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef short s16x2 __attribute__((ext_vector_type(2)));
__global__ void k(const float* in, int n, float scale, unsigned* o1, unsigned* o2, unsigned* o3) {
  int i = threadIdx.x;
  if (i >= n) return;
  float x = in[i];
  o1[i] = (unsigned)__builtin_amdgcn_cvt_pk_fp8_f32(x, 0.0f, 0, false) & 0xff;
  bf16x2 s = {(__bf16)x, (__bf16)0.0f};
  s16x2 old = {0, 0};
  s16x2 r = __builtin_amdgcn_cvt_scalef32_pk_fp8_bf16(old, s, scale, false);
  o2[i] = (unsigned)(unsigned short)r[0] & 0xff;
  o3[i] = (unsigned)__builtin_amdgcn_cvt_pk_fp8_f32(__builtin_amdgcn_fmed3f(x, -448.f, 448.f), 0.0f, 0, false) & 0xff;
}
static float e4m3(unsigned char v) { int s = v >> 7, e = (v >> 3) & 15, m = v & 7; float r; if (e == 15 && m == 7) r = NAN; else if (e == 0) r = ldexpf((float)m, -9); else r = ldexpf(1.0f + m / 8.0f, e - 7); return s ? -r : r; }
int main() {
  std::vector<float> in = {0.f, 1.f, 1.0625f, 1.1f, 1.125f, 1.1875f, -3.3f, 17.f, 18.f, 19.f, 208.f, 216.f, 440.f, 448.f, 460.f, 464.f, 480.f, 500.f, 1000.f, -1000.f, 0.001f, 0.002f, 0.003f, 0.0009765625f, 1e-5f, 30000.f};
  int n = in.size();
  float* d; unsigned *o1, *o2, *o3; hipMalloc(&d, n * 4); hipMalloc(&o1, n * 4); hipMalloc(&o2, n * 4); hipMalloc(&o3, n * 4);
  hipMemcpy(d, in.data(), n * 4, hipMemcpyHostToDevice);
  for (float scale : {1.0f, 4.0f, 0.25f}) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, n, scale, o1, o2, o3);
    std::vector<unsigned> a(n), b(n), c(n);
    hipMemcpy(a.data(), o1, n * 4, hipMemcpyDeviceToHost); hipMemcpy(b.data(), o2, n * 4, hipMemcpyDeviceToHost); hipMemcpy(c.data(), o3, n * 4, hipMemcpyDeviceToHost);
    printf("scale=%g\n", scale);
    for (int i = 0; i < n; ++i) printf("  x=%-12g cvt_pk_fp8_f32 -> 0x%02x (%g)   clamped -> 0x%02x (%g)   scalef32_bf16 -> 0x%02x (%g)\n", in[i], a[i], e4m3(a[i]), c[i], e4m3(c[i]), b[i], e4m3(b[i]));
  }
  return 0;
}
