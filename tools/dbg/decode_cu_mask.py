#!/usr/bin/env python3
"""Decode loop (large-v3, 128 forced tokens) on a stream confined to the first N compute units of the mask, nothing else running:
device-loop ms per call for 16 streams and for one.  (Round 4 observation: alone on 160 masked CUs the loop is FASTER than on all
256 - 173.1 vs 180.7 ms.)  python tools/dbg/decode_cu_mask.py [--cus 256,192,160,128,96,80]"""
import argparse, os, sys
os.environ["THEWHISPER_DECODE_CUS"] = "0"   # the engine's own default mask off: this tool sets the stream itself
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import bench
from thewhisper_amd.engine import WhisperEngine
from thewhisper_amd.overlap import masked_stream, _hiplib

ap = argparse.ArgumentParser()
ap.add_argument("--cus", default="256,192,160,128,96,80")
ap.add_argument("--streams", default="16,1")
ap.add_argument("--chunk-s", type=int, default=10)
args = ap.parse_args()
dev = torch.device("cuda", 0)
dims = bench.DIMS["large-v3"]
T = 50 * args.chunk_s
total = torch.cuda.get_device_properties(dev).multi_processor_count
for B in [int(x) for x in args.streams.split(",")]:
    eng = WhisperEngine(dims, T, max_batch=B, dtype="bf16", alignment_heads=bench.alignment_heads(dims), use_graph=True)
    eng.load_state_dict(bench.random_state_dict(dims, dev, 0))
    pcm = torch.randn((B, T * 320), device=dev) * 0.1
    prompt = np.tile(np.array([[50258, 50259, 50360]], dtype=np.int32), (B, 1))
    for n in [int(x) for x in args.cus.split(",")]:
        st = None
        if n < total:
            st = masked_stream(0, n, total, 0)
            eng.raw_stream = st
        else:
            eng.raw_stream = None
        best = 1e9
        for it in range(4):
            mel = eng.logmel(pcm); eng.encode(mel); eng.cross_kv(B)
            eng.generate_greedy(prompt, max_new_tokens=128, min_new_tokens=128, timestamps=True, want_alignment=True)
            tm = eng.last_timings()
            if it:
                best = min(best, tm["greedy_ms"])
        print(f"streams={B} decode CUs={n}: device loop {best:.2f} ms = {best / tm['decode_steps']:.4f} ms per step", flush=True)
        eng.raw_stream = None
        if st is not None:
            _hiplib().stream_destroy(st)
    eng.close(); del eng
    torch.cuda.empty_cache()
