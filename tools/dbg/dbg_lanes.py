"""16 streams per GPU as L independent lanes of 16/L streams, each lane = EncoderOverlap pipeline on its own slice of the CUs."""
import os, sys, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import bench
from thewhisper_amd.engine import WhisperEngine
from thewhisper_amd.overlap import EncoderOverlap
dims = bench.DIMS["large-v3"]; dev = torch.device("cuda", 0)
sd = bench.random_state_dict(dims, dev, 0)
L = int(os.environ.get("LANES", "2")); ENC = int(os.environ.get("ENC_CUS", "32")); NB = int(os.environ.get("BATCHES", "5"))
Bs = 16 // L
lanes = []
for l in range(L):
    engs = []
    for _ in range(2):
        e = WhisperEngine(dims, 500, max_batch=Bs, dtype="bf16", alignment_heads=bench.alignment_heads(dims), use_graph=True); e.load_state_dict(sd); engs.append(e)
    lanes.append(EncoderOverlap(engs, encoder_cus=ENC, cu_range=(l * 256 // L, (l + 1) * 256 // L)))
pcm = torch.randn((Bs, 160000), device=dev) * 0.1
prompt = np.tile(np.array([[50258, 50259, 50360]], dtype=np.int32), (Bs, 1))
def enc(e, pc):
    mel = e.logmel(pc); e.encode(mel); e.cross_kv(Bs)
    return mel
def dec(e, pc, _):
    out = e.generate_greedy(prompt, max_new_tokens=128, min_new_tokens=128, timestamps=True, want_alignment=True)
    e.token_timestamps(Bs, 3, out["length"], [1000] * Bs)
    return e.last_timings()["greedy_ms"]
def work(ov, n, res): res.append(ov.run([pcm] * n, enc, dec))
def go(n):
    res = []; th = [threading.Thread(target=work, args=(ov, n, res)) for ov in lanes]
    t0 = time.perf_counter(); [t.start() for t in th]; [t.join() for t in th]; torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3, res
go(2)
ms, res = go(NB)
print(f"lanes={L} x {Bs} streams, enc_cus={ENC}: {ms:.1f} ms for {NB} batches of 16 streams -> {16*128*NB/ms*1e3:.0f} tok/s; greedy_ms {[round(x) for x in res[0]]}")
