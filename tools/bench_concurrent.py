#!/usr/bin/env python3
"""Hypothesis test: the decode step is launch/latency-bound, so G independent contexts (each B/G streams, own HIP stream,
own host thread) should overlap.  Compares 1x16, 2x8, 4x4 streams on one MI355X."""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from thewhisper_amd.engine import WhisperEngine

LAYERS = int(os.environ.get("LAYERS", "8"))
TOK = 64
dims = dict(bench.DIMS["large-v3"], enc_layers=1, dec_layers=LAYERS)
dev = torch.device("cuda", 0)
sd = bench.random_state_dict(dims, dev, 0)
pcm = torch.randn((16, 160000), device=dev) * 0.1

def make(bs):
    e = WhisperEngine(dims, 500, max_batch=bs, dtype="bf16", alignment_heads=bench.alignment_heads(dims), use_graph=True)
    e.load_state_dict(sd)
    return e

for groups in (1, 2, 4):
    bs = 16 // groups
    engs = [make(bs) for _ in range(groups)]
    prompt = np.tile(np.array([[50258, 50259, 50360]], dtype=np.int32), (bs, 1))
    for e in engs:
        e.encode(e.logmel(pcm[:bs])); e.cross_kv(bs)
        s_ = torch.cuda.Stream(device=dev)
        with torch.cuda.stream(s_):   # warm-up on a non-default stream too (graph capture)
            e.generate_greedy(prompt, max_new_tokens=TOK, min_new_tokens=TOK, timestamps=True, want_alignment=True)
    torch.cuda.synchronize()
    def work(e):
        # every chain on ITS OWN HIP stream (a thread's torch "current stream" is the default stream otherwise, which would
        # serialise the groups on one queue and say nothing about concurrency)
        with torch.cuda.stream(torch.cuda.Stream(device=dev)):
            for _ in range(3):
                e.generate_greedy(prompt, max_new_tokens=TOK, min_new_tokens=TOK, timestamps=True, want_alignment=True)
    t0 = time.perf_counter()
    th = [threading.Thread(target=work, args=(e,)) for e in engs]
    [t.start() for t in th]; [t.join() for t in th]
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    steps = 3 * (TOK + 2)
    print(f"groups={groups} x B={bs}: {dt*1e3/steps:.4f} ms/step for 16 streams  ({16*3*TOK/dt:.0f} tok/s at {LAYERS} layers)", flush=True)
    for e in engs: e.close()
