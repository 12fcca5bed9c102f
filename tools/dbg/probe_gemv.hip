// Standalone probe of the decode projections: includes the product kernel source and replays a dependent chain of
// launches from a hipGraph (like one decode step does), printing microseconds per launch per shape.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 [-DTW_TIMING] tools/dbg/probe_gemv.hip -o tools/dbg/probe_gemv
#include "../../thewhisper_amd/csrc/k_decode.hip"
#include <algorithm>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

int main(int argc, char** argv) {
  const size_t pool_bytes = 768ull << 20;
  char* pool; CK(hipMalloc(&pool, pool_bytes)); CK(hipMemset(pool, 0x11, pool_bytes));
  bf16_t *x, *y, *res, *bias, *g, *b; float* logits; float *gw, *cb;
  CK(hipMalloc(&x, 64 * 5120 * 2)); CK(hipMalloc(&y, 64 * 5120 * 2)); CK(hipMalloc(&res, 64 * 5120 * 2));
  CK(hipMalloc(&bias, 52000 * 2)); CK(hipMalloc(&g, 1280 * 2)); CK(hipMalloc(&b, 1280 * 2)); CK(hipMalloc(&logits, 16 * 52000 * 4)); CK(hipMalloc(&gw, 52000 * 4)); CK(hipMalloc(&cb, 52000 * 4)); CK(hipMemset(gw, 0, 52000 * 4)); CK(hipMemset(cb, 0, 52000 * 4));
  CK(hipMemset(x, 0, 64 * 5120 * 2)); CK(hipMemset(y, 0, 64 * 5120 * 2)); CK(hipMemset(res, 0, 64 * 5120 * 2)); CK(hipMemset(bias, 0, 52000 * 2));
  CK(hipMemset(g, 0, 2560)); CK(hipMemset(b, 0, 2560));
  const int NL = 128;
  DecState* stt; CK(hipMalloc(&stt, NL * sizeof(DecState)));
  { std::vector<DecState> h(NL); for (int i = 0; i < NL; ++i) { h[i] = DecState{}; h[i].pos = i; } CK(hipMemcpy(stt, h.data(), NL * sizeof(DecState), hipMemcpyHostToDevice)); }
#ifdef TW_PROBE_TS
  unsigned long long* ts; const size_t ts_n = (size_t)NL * 1024 * 8; CK(hipMalloc(&ts, ts_n * 8));
  CK(hipMemcpyToSymbol(HIP_SYMBOL(g_probe_ts), &ts, sizeof(ts)));
#endif
  CK(init_decode_kernels());
  hipStream_t st;
  // PROBE_CUS=160: the launches run on a stream confined to that many compute units (what the decode loop gets in production,
  // thewhisper_amd/engine.py: THEWHISPER_DECODE_CUS = 160); default: the whole chip
  const int cus = getenv("PROBE_CUS") ? atoi(getenv("PROBE_CUS")) : 0;
  if (cus > 0) {
    unsigned mask[8] = {0};
    for (int i = 0; i < cus; ++i) mask[i / 32] |= 1u << (i % 32);
    CK(hipExtStreamCreateWithCUMask(&st, 8, mask));
  } else {
    CK(hipStreamCreate(&st));
  }
  // the decoder layer's projection launches as production lays them out (api.hip: retile): rows, K, rows per weight tile
  struct Shape { const char* name; int N, K; bool ln, resid, gelu; int tr; };
  const Shape shapes[] = {
      {"qkv+cq  5120x1280 LN    tr16", 5120, 1280, true, false, false, 16},
      {"o+comp  2560x1280 res   tr16", 2560, 1280, false, true, false, 16},
      {"o_cross 1280x1280 res   tr8 ", 1280, 1280, false, true, false, 8},
      {"fc1     5120x1280 LN G  tr16", 5120, 1280, true, false, true, 16},
      {"fc2     1280x5120 res   tr8 ", 1280, 5120, false, true, false, 8},
  };
  const int tr_narrow = getenv("PROBE_TR_NARROW") ? atoi(getenv("PROBE_TR_NARROW")) : 8;   // rows per tile of the N = 1280 launches
  std::vector<int> Bs = {16, 32, 64};
  if (getenv("PROBE_B")) { Bs.clear(); Bs.push_back(atoi(getenv("PROBE_B"))); }
  printf("# probe_gemv: CUs=%s TW_SK_CG_MODE=%s RED_STRIDE=%d CG_ORDER=%d CG_EPI_ALL=%d tr_narrow=%d\n", cus ? getenv("PROBE_CUS") : "all",
         getenv("TW_SK_CG_MODE") ? getenv("TW_SK_CG_MODE") : "(default)", TW_RED_STRIDE, TW_CG_ORDER, TW_CG_EPI_ALL, tr_narrow);
  for (int B : Bs) {
    double total_us = 0;
    for (const Shape& s : shapes) {
      const size_t wbytes = (size_t)s.N * s.K * 2;
      hipGraph_t gr; hipGraphExec_t ge;
      CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
      for (int i = 0; i < NL; ++i) {
        GemvArgs a{};
        a.x = (i & 1) ? y : x; a.ldx = s.K; a.W = pool + ((size_t)i * wbytes) % (pool_bytes - wbytes); a.bias = bias;
        a.N = s.N; a.K = s.K; a.B = B; a.gelu = s.gelu; a.y = (i & 1) ? x : y; a.ldy = s.N; a.tr = s.tr == 8 ? tr_narrow : s.tr;
        if (s.ln) { a.ln_gw = gw; a.ln_cb = cb; }
        if (s.resid) { a.res = res; a.ldres = s.N; }
        a.stt = stt + i;
        CK(launch_gemv(1, a, st));
      }
      CK(hipStreamEndCapture(st, &gr));
      CK(hipGraphInstantiate(&ge, gr, nullptr, nullptr, 0));
      CK(hipGraphLaunch(ge, st)); CK(hipStreamSynchronize(st));
#ifdef TW_PROBE_TS
      CK(hipMemset(ts, 0, ts_n * 8));
#endif
      hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
      CK(hipEventRecord(e0, st));
      const int R = 10;
      for (int r = 0; r < R; ++r) CK(hipGraphLaunch(ge, st));
      CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      const double us = ms * 1e3 / (R * NL);
      total_us += us;
      printf("B=%2d %s: %6.2f us/launch  (%.2f TB/s of weights)\n", B, s.name, us, wbytes / us * 1e-6);
#ifdef TW_PROBE_TS
      {
        std::vector<unsigned long long> h(ts_n);
        CK(hipMemcpy(h.data(), ts, ts_n * 8, hipMemcpyDeviceToHost));
        double span = 0, skew = 0, gap = 0, stage[8] = {0}; int nstage = 0, nblk = 0; unsigned long long prev_end = 0;
        for (int i = 0; i < NL; ++i) {
          unsigned long long s0 = ~0ull, s1 = 0, e1 = 0; double l[8] = {0}; int nb = 0;
          for (int bk = 0; bk < 1024; ++bk) {
            const unsigned long long* q = &h[((size_t)i * 1024 + bk) * 8];
            if (!q[0]) continue;
            ++nb; s0 = std::min(s0, q[0]); s1 = std::max(s1, q[0]);
            for (int k = 1; k < 8; ++k) if (q[k]) { l[k] += (double)(q[k] - q[0]); e1 = std::max(e1, q[k]); nstage = std::max(nstage, k); }
          }
          nblk = nb;
          span += (double)(e1 - s0); skew += (double)(s1 - s0);
          for (int k = 1; k < 8; ++k) stage[k] += l[k] / nb;
          if (i > 0) gap += (double)((long long)(s0 - prev_end));
          prev_end = e1;
        }
        printf("      blocks %d | span %.2f us, dispatch skew %.2f, gap %.2f | mean stamp offsets:", nblk, span / NL * 0.01, skew / NL * 0.01, gap / (NL - 1) * 0.01);
        for (int k = 1; k <= nstage; ++k) printf(" t%d=%.2f", k, stage[k] / NL * 0.01);
        printf("\n");
      }
#endif
      hipGraphExecDestroy(ge); hipGraphDestroy(gr);
    }
    printf("B=%2d sum of the five projection launches of a layer: %.2f us\n", B, total_us);
  }
  return 0;
}
