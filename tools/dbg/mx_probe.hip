// Probe of v_mfma_scale_f32_16x16x128_f8f6f4 (gfx950): operand layout and scale semantics, checked against a host emulation.
//   hipcc --offload-arch=gfx950 -O2 -std=c++17 tools/dbg/mx_probe.hip -o tools/dbg/mx_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v4f __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void k(const v8i* A, const v8i* B, const int* sa, const int* sb, v4f* D, v4f* D2) {
  const int l = threadIdx.x;
  v4f acc = {0.f, 0.f, 0.f, 0.f};
  D[l] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(A[l], B[l], acc, 0, 0, 0, sa[l], 0, sb[l]);
  // opsel = 1 on both: which byte of the scale register is used?
  D2[l] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(A[l], B[l], acc, 0, 0, 1, sa[l], 1, sb[l]);
}

static float e4m3(unsigned char v) {  // OCP e4m3fn
  const int s = v >> 7, e = (v >> 3) & 15, m = v & 7;
  float r;
  if (e == 15 && m == 7) r = NAN;
  else if (e == 0) r = ldexpf((float)m, -9);
  else r = ldexpf(1.0f + m / 8.0f, e - 7);
  return s ? -r : r;
}

int main() {
  std::vector<unsigned char> a(64 * 32), b(64 * 32);
  std::vector<int> sa(64), sb(64);
  srand(1);
  auto rnd8 = []() { unsigned char v; do { v = rand() & 0xff; } while ((v & 0x7f) == 0x7f); return v; };
  for (auto& x : a) x = rnd8();
  for (auto& x : b) x = rnd8();
  for (int l = 0; l < 64; ++l) {  // byte0 = per-lane exponent near 127, byte1 = another one
    sa[l] = (124 + (l * 7) % 7) | ((126 + l % 3) << 8);
    sb[l] = (125 + (l * 5) % 5) | ((127 + l % 2) << 8);
  }
  v8i *dA, *dB; int *dsa, *dsb; v4f *dD, *dD2;
  CK(hipMalloc(&dA, 2048)); CK(hipMalloc(&dB, 2048)); CK(hipMalloc(&dsa, 256)); CK(hipMalloc(&dsb, 256)); CK(hipMalloc(&dD, 1024)); CK(hipMalloc(&dD2, 1024));
  CK(hipMemcpy(dA, a.data(), 2048, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, b.data(), 2048, hipMemcpyHostToDevice));
  CK(hipMemcpy(dsa, sa.data(), 256, hipMemcpyHostToDevice)); CK(hipMemcpy(dsb, sb.data(), 256, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dsa, dsb, dD, dD2);
  CK(hipDeviceSynchronize());
  std::vector<float> D(256), D2(256);
  CK(hipMemcpy(D.data(), dD, 1024, hipMemcpyDeviceToHost)); CK(hipMemcpy(D2.data(), dD2, 1024, hipMemcpyDeviceToHost));
  // hypothesis: lane l = (row/col = l & 15, k-block = l >> 4) holds 32 consecutive k; its scale byte applies to that block;
  // D[i][j] in lane (j + 16*(i/4)), reg i%4
  for (int variant = 0; variant < 2; ++variant) {
    const std::vector<float>& Dv = variant ? D2 : D;
    double maxerr = 0, maxref = 0;
    for (int i = 0; i < 16; ++i)
      for (int j = 0; j < 16; ++j) {
        double ref = 0;
        for (int kb = 0; kb < 4; ++kb) {
          const int la = kb * 16 + i, lb = kb * 16 + j;
          const int ea = ((sa[la] >> (8 * variant)) & 0xff) - 127, eb = ((sb[lb] >> (8 * variant)) & 0xff) - 127;
          double part = 0;
          for (int e = 0; e < 32; ++e) part += (double)e4m3(a[la * 32 + e]) * (double)e4m3(b[lb * 32 + e]);
          ref += ldexp(part, ea + eb);
        }
        const float got = Dv[(j + 16 * (i / 4)) * 4 + (i % 4)];
        maxerr = fmax(maxerr, fabs(got - ref));
        maxref = fmax(maxref, fabs(ref));
      }
    printf("opsel=%d: max |got-ref| = %.6g (max |ref| = %.6g)\n", variant, maxerr, maxref);
  }
  return 0;
}
