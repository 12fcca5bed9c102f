import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import bench
from thewhisper_amd.engine import WhisperEngine
from thewhisper_amd.overlap import EncoderOverlap
dims = bench.DIMS["large-v3"]; dev = torch.device("cuda", 0)
sd = bench.random_state_dict(dims, dev, 0)
engs = []
for _ in range(2):
    e = WhisperEngine(dims, 500, max_batch=16, dtype="bf16", alignment_heads=bench.alignment_heads(dims), use_graph=True); e.load_state_dict(sd); engs.append(e)
pcm = torch.randn((16, 160000), device=dev) * 0.1
prompt = np.tile(np.array([[50258, 50259, 50360]], dtype=np.int32), (16, 1))
T0 = time.perf_counter(); log = []
def enc(e, pc):
    a = time.perf_counter() - T0
    mel = e.logmel(pc); e.encode(mel); e.cross_kv(16)
    log.append(("enc-submit", engs.index(e), round(a * 1e3, 1), round((time.perf_counter() - T0) * 1e3, 1)))
    return mel
def dec(e, pc, _):
    a = time.perf_counter() - T0
    out = e.generate_greedy(prompt, max_new_tokens=128, min_new_tokens=128, timestamps=True, want_alignment=True)
    b = time.perf_counter() - T0
    e.token_timestamps(16, 3, out["length"], [1000] * 16)
    log.append(("dec", engs.index(e), round(a * 1e3, 1), round(b * 1e3, 1), round((time.perf_counter() - T0) * 1e3, 1), e.last_timings()["greedy_ms"]))
ov = EncoderOverlap(engs, encoder_cus=int(os.environ.get("ENC_CUS", "32")))
ov.run([pcm] * 2, enc, dec); log.clear(); torch.cuda.synchronize(); T0 = time.perf_counter()
ov.run([pcm] * 5, enc, dec)
print("ENC_CUS", os.environ.get("ENC_CUS", "32"), "total ms for 5 batches:", round((time.perf_counter() - T0) * 1e3, 1), [l[-1] for l in log if l[0] == "dec"])
