#!/usr/bin/env python3
"""Decode-step microbenchmark (MI355X): large-v3 decoder dims with a reduced layer count, graph replay, ms/step for a few
batch sizes.  Used for A/B runs of kernel variants selected through TW_* environment variables."""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from thewhisper_amd.engine import WhisperEngine

ap = argparse.ArgumentParser()
ap.add_argument("--layers", type=int, default=8)
ap.add_argument("--batches", default="1,4,16")
ap.add_argument("--tokens", type=int, default=64)
ap.add_argument("--T", type=int, default=500)
ap.add_argument("--dtype", default="bf16")
args = ap.parse_args()
dims = dict(bench.DIMS["large-v3"], enc_layers=1, dec_layers=args.layers)
dev = torch.device("cuda", 0)
eng = WhisperEngine(dims, args.T, max_batch=max(int(x) for x in args.batches.split(',')), dtype=args.dtype, alignment_heads=bench.alignment_heads(dims), use_graph=True)
eng.load_state_dict(bench.random_state_dict(dims, dev, 0))
pcm = torch.randn((max(int(x) for x in args.batches.split(',')), args.T * 320), device=dev) * 0.1
for B in [int(x) for x in args.batches.split(",")]:
    eng.encode(eng.logmel(pcm[:B])); eng.cross_kv(B)
    prompt = np.tile(np.array([[50258, 50259, 50360]], dtype=np.int32), (B, 1))
    best = 1e9
    for it in range(4):
        eng.generate_greedy(prompt, max_new_tokens=args.tokens, min_new_tokens=args.tokens, timestamps=True, want_alignment=True)
        tm = eng.last_timings()
        best = min(best, tm["greedy_ms"] / tm["decode_steps"])
    print(f"B={B:2d} layers={args.layers} ms/step={best:.4f}  us/layer={best*1e3/args.layers:.1f}  env={ {k:v for k,v in os.environ.items() if k.startswith('TW_')} }", flush=True)
