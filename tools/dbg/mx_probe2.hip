// Which lane/byte of the scale operands of v_mfma_scale_f32_16x16x128_f8f6f4 applies to which (row, k-block)?
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v4f __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void k(const v8i* A, const v8i* B, const int* sa, const int* sb, v4f* D) {
  const int l = threadIdx.x;
  v4f acc = {0.f, 0.f, 0.f, 0.f};
  D[l] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(A[l], B[l], acc, 0, 0, 0, sa[l], 0, sb[l]);
}
static float e4m3(unsigned char v) { int s = v >> 7, e = (v >> 3) & 15, m = v & 7; float r; if (e == 15 && m == 7) r = NAN; else if (e == 0) r = ldexpf((float)m, -9); else r = ldexpf(1.0f + m / 8.0f, e - 7); return s ? -r : r; }
int main() {
  std::vector<unsigned char> a(64 * 32), b(64 * 32);
  srand(1);
  auto rnd8 = []() { unsigned char v; do { v = rand() & 0xff; } while ((v & 0x7f) == 0x7f || ((v >> 3) & 15) > 9); return v; };
  for (auto& x : a) x = rnd8();
  for (auto& x : b) x = rnd8();
  v8i *dA, *dB; int *dsa, *dsb; v4f* dD;
  CK(hipMalloc(&dA, 2048)); CK(hipMalloc(&dB, 2048)); CK(hipMalloc(&dsa, 256)); CK(hipMalloc(&dsb, 256)); CK(hipMalloc(&dD, 1024));
  CK(hipMemcpy(dA, a.data(), 2048, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, b.data(), 2048, hipMemcpyHostToDevice));
  // partial[i][j][kb] under the data layout already verified (lane = kb*16 + row/col, 32 consecutive k)
  static double part[16][16][4];
  for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) for (int kb = 0; kb < 4; ++kb) {
    double p = 0; for (int e = 0; e < 32; ++e) p += (double)e4m3(a[(kb * 16 + i) * 32 + e]) * e4m3(b[(kb * 16 + j) * 32 + e]);
    part[i][j][kb] = p;
  }
  // H4: a lane's 32 bytes are 4 chunks of 8; chunk c of lane (kb, row) holds true k = c*32 + kb*8 + e, so the scale block t
  // (32 consecutive k) is chunk t of the row's four lanes and its scale sits in lane t*16 + row
  static double part4[16][16][4];
  for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) for (int t = 0; t < 4; ++t) {
    double p = 0;
    for (int kb = 0; kb < 4; ++kb) for (int e = 0; e < 8; ++e) p += (double)e4m3(a[(kb * 16 + i) * 32 + t * 8 + e]) * e4m3(b[(kb * 16 + j) * 32 + t * 8 + e]);
    part4[i][j][t] = p;
  }
  auto run = [&](const std::vector<int>& sa, const std::vector<int>& sb, std::vector<float>& D) {
    hipMemcpy(dsa, sa.data(), 256, hipMemcpyHostToDevice); hipMemcpy(dsb, sb.data(), 256, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dsa, dsb, dD);
    hipDeviceSynchronize(); D.resize(256); hipMemcpy(D.data(), dD, 1024, hipMemcpyDeviceToHost);
  };
  std::vector<int> base(64, 127 | (127 << 8) | (127 << 16) | (127 << 24));
  std::vector<float> D0; run(base, base, D0);
  // sub-partials: sub[i][j][kb][c] = sum over the 8 bytes of chunk c of lane (kb, .)
  static double sub[16][16][4][4];
  for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) for (int kb = 0; kb < 4; ++kb) for (int c = 0; c < 4; ++c) {
    double p = 0; for (int e = 0; e < 8; ++e) p += (double)e4m3(a[(kb * 16 + i) * 32 + c * 8 + e]) * e4m3(b[(kb * 16 + j) * 32 + c * 8 + e]);
    sub[i][j][kb][c] = p;
  }
  for (int which = 0; which < 2; ++which)
    for (int L = 0; L < 64; ++L) {
      std::vector<int> s = base; s[L] = (s[L] & ~0xff) | 128;
      std::vector<float> D; if (which == 0) run(s, base, D); else run(base, s, D);
      int idx = -1; for (int q = 0; q < 16 && idx < 0; ++q) for (int o = 0; o < 16; ++o) { int i = which == 0 ? q : o, j = which == 0 ? o : q; if (D[(j + 16 * (i / 4)) * 4 + (i % 4)] != D0[(j + 16 * (i / 4)) * 4 + (i % 4)]) { idx = q; break; } }
      if (idx < 0) { printf("scale_%c lane %2d: no effect\n", which ? 'b' : 'a', L); continue; }
      // which set of sub-blocks (mask over 16 = kb*4+c) explains the change for every partner index?
      int best_mask = -1;
      for (int mask = 1; mask < 65536 && best_mask < 0; ++mask) {
        if (__builtin_popcount(mask) != 4) continue;
        double err = 0;
        for (int o = 0; o < 16; ++o) {
          int i = which == 0 ? idx : o, j = which == 0 ? o : idx;
          double want = 0; for (int q = 0; q < 16; ++q) if (mask >> q & 1) want += sub[i][j][q / 4][q % 4];
          err = fmax(err, fabs((double)D[(j + 16 * (i / 4)) * 4 + (i % 4)] - D0[(j + 16 * (i / 4)) * 4 + (i % 4)] - want));
        }
        if (err < 2e-3) best_mask = mask;
      }
      printf("scale_%c lane %2d -> %s %2d, sub-blocks (lane-group kb, chunk c):", which ? 'b' : 'a', L, which ? "col" : "row", idx);
      if (best_mask < 0) printf(" ?"); else for (int q = 0; q < 16; ++q) if (best_mask >> q & 1) printf(" (%d,%d)", q / 4, q % 4);
      printf("\n");
    }
  return 0;
}
