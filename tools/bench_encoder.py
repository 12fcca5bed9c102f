#!/usr/bin/env python3
"""Encoder-stage microbenchmark (MI355X): large-v3 encoder (32 layers) + cross-K/V projection, ms per call and achieved
TFLOP/s against SURVEY.md section 8d's algorithmic flops, for (T, B) pairs.  A/B runs of kernel variants via TW_* env."""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from thewhisper_amd.engine import WhisperEngine

ap = argparse.ArgumentParser()
ap.add_argument("--cases", default="500x16,1500x1,500x1,750x16")
ap.add_argument("--dec-layers", type=int, default=4)
ap.add_argument("--dtype", default="bf16")
args = ap.parse_args()
dev = torch.device("cuda", 0)
for case in args.cases.split(","):
    T, B = (int(x) for x in case.split("x"))
    dims = dict(bench.DIMS["large-v3"], dec_layers=args.dec_layers)
    eng = WhisperEngine(dims, T, max_batch=B, dtype=args.dtype, alignment_heads=[(0, 0)], use_graph=False)
    eng.load_state_dict(bench.random_state_dict(dims, dev, 0))
    pcm = torch.randn((B, T * 320), device=dev) * 0.1
    mel = eng.logmel(pcm)
    best_e, best_c = 1e9, 1e9
    for it in range(5):
        eng.encode(mel); eng.cross_kv(B)
        torch.cuda.synchronize()
        tm = eng.last_timings()
        best_e, best_c = min(best_e, tm["encode_ms"]), min(best_c, tm["cross_kv_ms"])
    d, L = 1280, 32
    flops_enc = B * (11796480 * T + L * T * (39321600 + 5120 * T))       # conv stem + encoder layers
    flops_ckv = B * args.dec_layers * T * 2 * 2 * d * d
    print(f"T={T} B={B} encode_ms={best_e:.3f} ({flops_enc / best_e / 1e9:.0f} TF/s = {flops_enc / best_e / 1e9 / 25:.1f} % of 2.5 PF)  "
          f"cross_kv_ms={best_c:.3f} ({flops_ckv / best_c / 1e9:.0f} TF/s)  env={ {k: v for k, v in os.environ.items() if k.startswith('TW_')} }", flush=True)
    eng.close()
    del eng
    torch.cuda.empty_cache()
if os.environ.get("TW_TORCH_REF"):
    # comparison only (never on the product path): what the vendor library reaches on the encoder's GEMM shapes
    for (M, N, K) in [(8000, 5120, 1280), (8000, 1280, 5120), (8000, 3840, 1280), (8000, 1280, 1280), (1500, 5120, 1280), (500, 5120, 1280)]:
        a = torch.randn((M, K), device=dev, dtype=torch.bfloat16); w = torch.randn((N, K), device=dev, dtype=torch.bfloat16)
        for _ in range(3): torch.matmul(a, w.t())
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): torch.matmul(a, w.t())
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        print(f"torch.matmul (hipBLASLt, comparison only) M={M} N={N} K={K}: {ms*1e3:.1f} us = {2*M*N*K/ms/1e9:.0f} TF/s", flush=True)
