#!/usr/bin/env python3
"""Where does the host time of one batched pipeline call go?  cProfile of AMDWhisperBackend.transcribe_many over 16 ten-second
buffers (large-v3 dims, random weights) on the MI355X."""
import cProfile, io, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from transformers import WhisperFeatureExtractor
import bench
from thewhisper_amd import AMDWhisperBackend, ASRPipeline, synthetic
from thewhisper_amd.engine import WhisperEngine

dims = bench.DIMS["large-v3"]; B = 16; heads = bench.alignment_heads(dims)
dev = torch.device("cuda", 0)
eng = WhisperEngine(dims, 500, max_batch=B, dtype="bf16", alignment_heads=heads)
eng.load_state_dict(bench.random_state_dict(dims, dev, 0))
model = synthetic.skeleton_model(dims, device="cuda:0", dtype=torch.bfloat16, alignment_heads=heads)
pipe = ASRPipeline(model, feature_extractor=WhisperFeatureExtractor(feature_size=128, chunk_length=10), tokenizer=synthetic.build_tokenizer(dims["vocab"]),
                   chunk_length_s=10, device="cuda:0", torch_dtype=torch.bfloat16, batch_size=B, engine=eng)
be = AMDWhisperBackend(None, chunk_length_s=10, asr_pipeline=pipe)
rng = np.random.default_rng(7)
clips = [(rng.standard_normal(160000) * 0.1).clip(-1, 1).astype(np.float32) for _ in range(B)]
reqs = [(c, 0.0, 16000) for c in clips]
be.transcribe_many(reqs, batch_size=B)
t_eng = {"n": 0, "t": 0.0}
for name in ("generate_greedy", "encode", "cross_kv", "logmel", "token_timestamps"):
    f = getattr(eng, name)
    def wrap(*a, _f=f, **k):
        t0 = time.perf_counter(); r = _f(*a, **k); torch.cuda.synchronize(); t_eng["t"] += time.perf_counter() - t0; t_eng["n"] += 1; return r
    setattr(eng, name, wrap)
pr = cProfile.Profile()
t0 = time.perf_counter()
pr.enable(); be.transcribe_many(reqs, batch_size=B); pr.disable()
wall = time.perf_counter() - t0
print(f"wall {wall*1e3:.1f} ms, inside engine calls (synchronised) {t_eng['t']*1e3:.1f} ms over {t_eng['n']} calls")
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(45); print(s.getvalue()[:9000])
