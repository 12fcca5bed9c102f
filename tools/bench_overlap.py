#!/usr/bin/env python3
"""Does the encoder of the NEXT batch overlap with the decode loop of the CURRENT one (two contexts, two HIP streams, two host
threads)?  Prints the wall time of N decode calls alone, N encode calls alone, and both together."""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from thewhisper_amd.engine import WhisperEngine

dims = bench.DIMS["large-v3"]
dev = torch.device("cuda", 0)
sd = bench.random_state_dict(dims, dev, 0)
pcm = torch.randn((16, 160000), device=dev) * 0.1
A = WhisperEngine(dims, 500, max_batch=16, dtype="bf16", alignment_heads=bench.alignment_heads(dims), use_graph=True)
A.load_state_dict(sd)
Bn = WhisperEngine(dims, 500, max_batch=16, dtype="bf16", alignment_heads=bench.alignment_heads(dims), use_graph=True)
Bn.load_state_dict(sd)
prompt = np.tile(np.array([[50258, 50259, 50360]], dtype=np.int32), (16, 1))
sA, sB = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
ENC_CUS = int(os.environ.get("ENC_CUS", "0"))
if ENC_CUS:  # confine the encoder context to ENC_CUS compute units and the decoder context to the rest
    import ctypes as C
    hip = C.CDLL("libamdhip64.so")
    def masked(lo, hi):
        m = (C.c_uint32 * 8)(*[0] * 8)
        for i in range(lo, hi):
            m[i // 32] |= 1 << (i % 32)
        st = C.c_void_p()
        rc = hip.hipExtStreamCreateWithCUMask(C.byref(st), 8, m)
        assert rc == 0, rc
        return st.value
    A.raw_stream = masked(0, 256 - ENC_CUS)
    Bn.raw_stream = masked(256 - ENC_CUS, 256)
    hip.hipStreamSynchronize.argtypes = [C.c_void_p]
N = 3

def dec():
    with torch.cuda.stream(sA):
        for _ in range(N):
            A.generate_greedy(prompt, max_new_tokens=128, min_new_tokens=128, timestamps=True, want_alignment=True)
        sA.synchronize()
        if ENC_CUS: hip.hipStreamSynchronize(A.raw_stream)

def enc():
    with torch.cuda.stream(sB):
        for _ in range(N * 4):
            Bn.encode(Bn.logmel(pcm)); Bn.cross_kv(16)
        sB.synchronize()
        if ENC_CUS: hip.hipStreamSynchronize(Bn.raw_stream)

with torch.cuda.stream(sA):
    A.encode(A.logmel(pcm)); A.cross_kv(16); A.generate_greedy(prompt, max_new_tokens=8, min_new_tokens=8, timestamps=True, want_alignment=True)
with torch.cuda.stream(sB):
    Bn.encode(Bn.logmel(pcm)); Bn.cross_kv(16)
torch.cuda.synchronize()
def timed(fns):
    th = [threading.Thread(target=f) for f in fns]
    t0 = time.perf_counter(); [t.start() for t in th]; [t.join() for t in th]; torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3
td = timed([dec]); te = timed([enc]); tb = timed([dec, enc])
print(f"decode x{N}: {td:.1f} ms   encode x{N*4}: {te:.1f} ms   both concurrently: {tb:.1f} ms   (sum {td+te:.1f}, max {max(td,te):.1f})")
