// Stand-alone check of the MXFP8 projection path (quant_mx8_kernel + skinny_mfma_kernel<W8>) against a host emulation.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/dbg/mx8_gemv_probe.hip -o tools/dbg/mx8_gemv_probe
#include "../../thewhisper_amd/csrc/k_decode.hip"
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
static float bf16r(float f) { unsigned u; memcpy(&u, &f, 4); u = (u + 0x7fff + ((u >> 16) & 1)) & 0xffff0000u; memcpy(&f, &u, 4); return f; }
static unsigned short bf16bits(float f) { unsigned u; f = bf16r(f); memcpy(&u, &f, 4); return (unsigned short)(u >> 16); }
static double e4m3_rne(double v) { double a = fabs(v); if (a == 0) return 0; double e = floor(log2(fmax(a, ldexp(1.0, -6)))); double step = ldexp(1.0, (int)e - 3); return copysign(nearbyint(a / step) * step, v); }
// quantise-dequantise K values of one row in the kernel's block structure
static void qdq(const float* x, int K, float* out) {
  for (int s = 0; s < K / 128; ++s)
    for (int h = 0; h < 2; ++h)
      for (int u = 0; u < 2; ++u) {
        float amax = 0;
        for (int mm = 0; mm < 2; ++mm) for (int kk = 0; kk < 2; ++kk) for (int e = 0; e < 8; ++e) amax = fmaxf(amax, fabsf(x[s * 128 + (2 * h + mm) * 32 + (2 * u + kk) * 8 + e]));
        int Eb = amax > 0 ? (int)floor(log2(amax)) + 127 : 0;
        int sb = Eb - 7 > 1 ? Eb - 7 : 1;
        double X = ldexp(1.0, sb - 127);
        for (int mm = 0; mm < 2; ++mm) for (int kk = 0; kk < 2; ++kk) for (int e = 0; e < 8; ++e) { int k = s * 128 + (2 * h + mm) * 32 + (2 * u + kk) * 8 + e; out[k] = (float)(e4m3_rne(x[k] / X) * X); }
      }
}
__global__ void quant_blocks(const u32x4_t* in, v8i_t* out, int* sc, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  u32x4_t xv[4] = {in[i * 4], in[i * 4 + 1], in[i * 4 + 2], in[i * 4 + 3]};
  int sb;
  out[i] = sk_quant_mx8(xv, sb);
  sc[i] = sb;
}
static float e4m3dec(unsigned char v) { int s = v >> 7, e = (v >> 3) & 15, m = v & 7; float r; if (e == 15 && m == 7) r = NAN; else if (e == 0) r = ldexpf((float)m, -9); else r = ldexpf(1.0f + m / 8.0f, e - 7); return s ? -r : r; }
static int check_quant() {
  const int n = 4096;
  std::vector<float> v(n * 32);
  srand(7);
  for (int i = 0; i < n; ++i) { float amp = ldexpf(1.f, (rand() % 24) - 12); for (int k = 0; k < 32; ++k) v[i * 32 + k] = bf16r((rand() / (float)RAND_MAX - 0.5f) * amp); }
  std::vector<unsigned short> vb(n * 32); for (int i = 0; i < n * 32; ++i) vb[i] = bf16bits(v[i]);
  u32x4_t* din; v8i_t* dout; int* dsc;
  CK(hipMalloc(&din, n * 64)); CK(hipMalloc(&dout, n * 32)); CK(hipMalloc(&dsc, n * 4));
  CK(hipMemcpy(din, vb.data(), n * 64, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(quant_blocks, dim3(n / 256), dim3(256), 0, 0, din, dout, dsc, n);
  CK(hipDeviceSynchronize());
  std::vector<unsigned char> q(n * 32); std::vector<int> sc(n);
  CK(hipMemcpy(q.data(), dout, n * 32, hipMemcpyDeviceToHost)); CK(hipMemcpy(sc.data(), dsc, n * 4, hipMemcpyDeviceToHost));
  int bad = 0;
  for (int i = 0; i < n; ++i) {
    float amax = 0; for (int k = 0; k < 32; ++k) amax = fmaxf(amax, fabsf(v[i * 32 + k]));
    int Eb = amax > 0 ? (int)floor(log2(amax)) + 127 : 0; int sb = Eb - 7 > 1 ? Eb - 7 : 1; double X = ldexp(1.0, sb - 127);
    for (int k = 0; k < 32; ++k) {
      const double want = e4m3_rne(v[i * 32 + k] / X); const float got = e4m3dec(q[i * 32 + k]);
      if ((sc[i] != sb || got != (float)want) && bad++ < 10) printf("  quant mismatch block %d elem %d: v=%g amax=%g sb %d/%d got %g want %g\n", i, k, v[i * 32 + k], amax, sc[i], sb, got, want);
    }
  }
  printf("quant check: %d mismatches of %d\n", bad, n * 32);
  return 0;
}
int main() {

  const int N = 48, K = 256, B = 16;
  std::vector<float> W(N * K), x(B * K);
  srand(3);
  for (auto& v : W) v = bf16r((rand() / (float)RAND_MAX - 0.5f) * 0.2f);
  for (int b = 0; b < B; ++b) for (int k = 0; k < K; ++k) x[b * K + k] = bf16r((rand() / (float)RAND_MAX - 0.5f) * (b + 1) * 3.f);
  std::vector<unsigned short> Wb(N * K), xt(16 * K, 0);
  for (int i = 0; i < N * K; ++i) Wb[i] = bf16bits(W[i]);
  for (int b = 0; b < B; ++b) for (int k = 0; k < K; ++k) xt[((k / 32) * 64 + ((k / 8) & 3) * 16 + b) * 8 + (k % 8)] = bf16bits(x[b * K + k]);
  bf16_t *dW, *dx, *dy; unsigned char *dq, *ds;
  CK(hipMalloc(&dW, N * K * 2)); CK(hipMalloc(&dx, 16 * K * 2)); CK(hipMalloc(&dy, B * N * 2)); CK(hipMalloc(&dq, N * K)); CK(hipMalloc(&ds, N * K / 32));
  CK(hipMemcpy(dW, Wb.data(), N * K * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(dx, xt.data(), 16 * K * 2, hipMemcpyHostToDevice));
  CK(launch_quant_mx8(dW, dq, ds, N, K, 0));
  GemvArgs a{};
  a.x = dx; a.W = dq; a.wscale = ds; a.N = N; a.K = K; a.B = B; a.y = dy; a.ldy = N;
  CK(launch_gemv(1, a, 0));
  CK(hipDeviceSynchronize());
  std::vector<unsigned short> yb(B * N); CK(hipMemcpy(yb.data(), dy, B * N * 2, hipMemcpyDeviceToHost));
  std::vector<unsigned char> hs(N * K / 32); CK(hipMemcpy(hs.data(), ds, hs.size(), hipMemcpyDeviceToHost));
  printf("first scale bytes: %d %d %d %d\n", hs[0], hs[1], hs[16], hs[17]);
  std::vector<float> Wq(N * K), xq(B * K);
  for (int n = 0; n < N; ++n) qdq(&W[n * K], K, &Wq[n * K]);
  for (int b = 0; b < B; ++b) qdq(&x[b * K], K, &xq[b * K]);
  double maxerr = 0, maxref = 0, maxerr_unq = 0; int nbad = 0; int badrow[16] = {0};
  for (int b = 0; b < B; ++b)
    for (int n = 0; n < N; ++n) {
      double r = 0, ru = 0;
      for (int k = 0; k < K; ++k) { r += (double)Wq[n * K + k] * xq[b * K + k]; ru += (double)W[n * K + k] * x[b * K + k]; }
      unsigned u = (unsigned)yb[b * N + n] << 16; float got; memcpy(&got, &u, 4);
      if (fabs(got - r) > 0.05 * fabs(r) + 0.05) ++badrow[b];
      if (fabs(got - r) > 0.05 * fabs(r) + 0.05 && nbad++ < 3) printf("  BAD y[%d][%d] = %9.5f emulated %9.5f exact %9.5f\n", b, n, got, r, ru);
      maxerr = fmax(maxerr, fabs(got - r)); maxref = fmax(maxref, fabs(r)); maxerr_unq = fmax(maxerr_unq, fabs(got - ru));
      if (b < 2 && n < 4) printf("  y[%d][%d] = %9.5f  emulated %9.5f  exact %9.5f\n", b, n, got, r, ru);
    }
  for (int b = 0; b < B; ++b) { float am = 0; for (int k = 0; k < K; ++k) am = fmaxf(am, fabsf(x[b * K + k])); printf("  stream %2d: %d bad of %d, amax %.3f\n", b, badrow[b], N, am); }
  printf("max |got - emulated| = %.5g, max |got - exact| = %.5g, max |ref| = %.5g\n", maxerr, maxerr_unq, maxref);
  // ---- folded-LayerNorm + fp32 output flavour (the logits projection) ----
  {
    std::vector<float> gw(N), cb(N);
    for (int n = 0; n < N; ++n) { gw[n] = (rand() / (float)RAND_MAX - 0.5f); cb[n] = (rand() / (float)RAND_MAX - 0.5f); }
    float *dgw, *dcb, *dyf;
    CK(hipMalloc(&dgw, N * 4)); CK(hipMalloc(&dcb, N * 4)); CK(hipMalloc(&dyf, B * N * 4));
    CK(hipMemcpy(dgw, gw.data(), N * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dcb, cb.data(), N * 4, hipMemcpyHostToDevice));
    GemvArgs a2{};
    a2.x = dx; a2.W = dq; a2.wscale = ds; a2.N = N; a2.K = K; a2.B = B; a2.y_f32 = dyf; a2.ln_gw = dgw; a2.ln_cb = dcb;
    CK(launch_gemv(1, a2, 0));
    CK(hipDeviceSynchronize());
    std::vector<float> yf(B * N); CK(hipMemcpy(yf.data(), dyf, B * N * 4, hipMemcpyDeviceToHost));
    double me = 0, mr = 0;
    for (int b = 0; b < B; ++b) {
      double sm = 0, sq = 0; for (int k = 0; k < K; ++k) { sm += x[b * K + k]; sq += (double)x[b * K + k] * x[b * K + k]; }
      const double mean = sm / K, rstd = 1.0 / sqrt(fmax(sq / K - mean * mean, 0.0) + 1e-5);
      for (int n = 0; n < N; ++n) {
        double r = 0; for (int k = 0; k < K; ++k) r += (double)Wq[n * K + k] * xq[b * K + k];
        const double want = rstd * (r - mean * gw[n]) + cb[n];
        me = fmax(me, fabs(yf[b * N + n] - want)); mr = fmax(mr, fabs(want));
        if (b == 3 && n < 3) printf("  LN y[%d][%d] = %9.5f emulated %9.5f\n", b, n, yf[b * N + n], want);
      }
    }
    printf("LN + f32 flavour: max |got - emulated| = %.5g (max |ref| %.5g)\n", me, mr);
  }
  // ---- folded-LayerNorm + fp32 output flavour (the logits projection) ----
  {
    std::vector<float> gw(N), cb(N);
    for (int n = 0; n < N; ++n) { gw[n] = (rand() / (float)RAND_MAX - 0.5f); cb[n] = (rand() / (float)RAND_MAX - 0.5f); }
    float *dgw, *dcb, *dyf;
    CK(hipMalloc(&dgw, N * 4)); CK(hipMalloc(&dcb, N * 4)); CK(hipMalloc(&dyf, B * N * 4));
    CK(hipMemcpy(dgw, gw.data(), N * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dcb, cb.data(), N * 4, hipMemcpyHostToDevice));
    GemvArgs a2{};
    a2.x = dx; a2.W = dq; a2.wscale = ds; a2.N = N; a2.K = K; a2.B = B; a2.y_f32 = dyf; a2.ln_gw = dgw; a2.ln_cb = dcb;
    CK(launch_gemv(1, a2, 0));
    CK(hipDeviceSynchronize());
    std::vector<float> yf(B * N); CK(hipMemcpy(yf.data(), dyf, B * N * 4, hipMemcpyDeviceToHost));
    double me = 0, mr = 0;
    for (int b = 0; b < B; ++b) {
      double sm = 0, sq = 0; for (int k = 0; k < K; ++k) { sm += x[b * K + k]; sq += (double)x[b * K + k] * x[b * K + k]; }
      const double mean = sm / K, rstd = 1.0 / sqrt(fmax(sq / K - mean * mean, 0.0) + 1e-5);
      for (int n = 0; n < N; ++n) {
        double r = 0; for (int k = 0; k < K; ++k) r += (double)Wq[n * K + k] * xq[b * K + k];
        const double want = rstd * (r - mean * gw[n]) + cb[n];
        me = fmax(me, fabs(yf[b * N + n] - want)); mr = fmax(mr, fabs(want));
        if (b == 3 && n < 3) printf("  LN y[%d][%d] = %9.5f emulated %9.5f\n", b, n, yf[b * N + n], want);
      }
    }
    printf("LN + f32 flavour: max |got - emulated| = %.5g (max |ref| %.5g)\n", me, mr);
  }
  return 0;
}
