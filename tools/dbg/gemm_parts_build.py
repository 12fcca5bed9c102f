#!/usr/bin/env python3
"""What bounds the large-M GEMM: builds of libthewhisper with PARTS of gemm_wreg_kernel compiled out (round 4, profiles/r04_gemm_parts.txt).

    python tools/dbg/gemm_parts_build.py [bits ...]        # default: 0 1 2 14 15   ->  thewhisper_amd/lib/dbg/libtw_gemm_dbg<bits>.so
    THEWHISPER_LIB=.../libtw_gemm_dbg1.so python tools/bench_encoder.py --cases 500x16     (under rocprofv3 + tools/trace_by_shape.py)

bits: 1 no epilogue (the accumulators are folded into one compare so that the MFMAs stay), 2 weight fragments loaded once, 4 no
activation DMA after the prologue, 8 no LDS fragment reads.  4 alone or 4 + 8 without 2 HANG (the wait counts of the loop expect the
requests): use 2, 14 (= 2 + 4 + 8) and 15.  Results are wrong by construction - only the time is read.  The kernel source is patched in a
scratch copy; nothing of this is in the product."""
import os, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
SRC = os.path.join(ROOT, "thewhisper_amd", "csrc", "k_gemm.hip")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc", "-Wno-unused-result", "-mllvm", "-amdgpu-kernarg-preload-count=16"]


def patched() -> str:
    s = open(SRC).read()

    def rep(old, new):
        nonlocal s
        assert s.count(old) == 1, old[:60]
        s = s.replace(old, new)

    rep("""    tw_barrier_only();
    load_w(nxt, min(kt + 1, nk - 1));
    {
      int nb = buf + ST - 1;
      if (nb >= ST) nb -= ST;
      issue_a(min(kt + ST - 1, nk - 1), nb);
    }""", """    tw_barrier_only();
#if !(TW_GEMM_DBG & 2)
    load_w(nxt, min(kt + 1, nk - 1));
#endif
#if !(TW_GEMM_DBG & 4)
    {
      int nb = buf + ST - 1;
      if (nb >= ST) nb -= ST;
      issue_a(min(kt + ST - 1, nk - 1), nb);
    }
#endif""")
    rep("""      tw_static_for<0, MT>([&](auto bc) { tw_lds_read<decltype(bc)::value * 2048>(af[decltype(bc)::value], a_addr[kk] + stage_off); });
      // fragments are consumed in arrival order: MFMAs on af[b] start while af[b+1..] are still on their way
      tw_static_for<0, MT>([&](auto bc) {
        constexpr int b = decltype(bc)::value;
        tw_lds_ready<MT - 1 - b>(af[b]);""", """#if !(TW_GEMM_DBG & 8)
      tw_static_for<0, MT>([&](auto bc) { tw_lds_read<decltype(bc)::value * 2048>(af[decltype(bc)::value], a_addr[kk] + stage_off); });
#else
      tw_static_for<0, MT>([&](auto bc) { af[decltype(bc)::value] = u32x4_t{(unsigned)kt, (unsigned)lane, 0x3f803f80u, stage_off}; tw_tie(af[decltype(bc)::value]); });
#endif
      tw_static_for<0, MT>([&](auto bc) {
        constexpr int b = decltype(bc)::value;
#if !(TW_GEMM_DBG & 8)
        tw_lds_ready<MT - 1 - b>(af[b]);
#endif""")
    rep("""  issue_a(0, 0);
  load_w(w0, 0);
  if (ST == 3) issue_a(min(1, nk - 1), 1);""", """  issue_a(0, 0);
  load_w(w0, 0);
#if (TW_GEMM_DBG & 2)
  load_w(w1, min(1, nk - 1));
#endif
  if (ST == 3) issue_a(min(1, nk - 1), 1);""")
    a = s.index("  if constexpr (sizeof(T) == 2) {\n    // rows contiguous over this wavefront's 64 columns")
    tail = "  else gemm_epilogue<T, NT, MT>(acc, m0, n0, M, N, ep, fr, fq, ep.stats_in ? ln_rows : nullptr);\n"
    b = s.index(tail, a) + len(tail)
    s = s[:a] + """#if (TW_GEMM_DBG & 1)
  {
    float sacc = 0.f;
#pragma unroll
    for (int a = 0; a < NT; ++a)
#pragma unroll
      for (int b = 0; b < MT; ++b) sacc += acc[a][b][0] + acc[a][b][1] + acc[a][b][2] + acc[a][b][3];
    if (sacc == 12345.678f) reinterpret_cast<float*>(ep.out)[tid] = sacc;
  }
#else
""" + s[a:b] + "#endif\n" + s[b:]
    return "#ifndef TW_GEMM_DBG\n#define TW_GEMM_DBG 0\n#endif\n" + s


def main():
    bits = [int(x) for x in sys.argv[1:]] or [0, 1, 2, 14, 15]
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    obj = os.path.join(ROOT, "thewhisper_amd", "lib", "obj")
    out = os.path.join(ROOT, "thewhisper_amd", "lib", "dbg")
    os.makedirs(out, exist_ok=True)
    others = [os.path.join(obj, f) for f in sorted(os.listdir(obj)) if f.endswith(".o") and f != "k_gemm.o"]
    assert others, "build the library first (python __graft_entry__.py)"
    with tempfile.TemporaryDirectory() as tmp:
        src = os.path.join(tmp, "k_gemm.hip")
        open(src, "w").write(patched())
        for b in bits:
            o = os.path.join(tmp, f"k_gemm_{b}.o")
            subprocess.check_call([hipcc, *FLAGS, f"-DTW_GEMM_DBG={b}", "-I", os.path.join(ROOT, "thewhisper_amd", "csrc"), "-c", src, "-o", o])
            lib = os.path.join(out, f"libtw_gemm_dbg{b}.so")
            subprocess.check_call([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib, o, *others])
            print(lib)


if __name__ == "__main__":
    main()
