#!/usr/bin/env python3
"""Round 5 diagnostic 3: the failing property (tests/test_gpu_parity.py::test_large_v3_maximum_context_properties: streams 17 and 63
repeat stream 0's audio -> same greedy ids) at smaller sizes, with the loop's switches."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import numpy as np, torch
import bench
from tests.util import PROMPT, clips
from thewhisper_amd.engine import WhisperEngine

layers, T, graph, new = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
dims = dict(bench.DIMS["large-v3"], enc_layers=int(os.environ.get("ENC_LAYERS", "2")), dec_layers=layers)
heads = bench.alignment_heads(dims)
eng = WhisperEngine(dims, T, max_batch=64, dtype="bf16", alignment_heads=heads, use_graph=bool(graph))
eng.load_state_dict(bench.random_state_dict(dims, torch.device("cuda", 0), seed=0))
kinds = ["speechlike", "noise", "sine", "speechlike"]
pcm = torch.from_numpy(clips(T * 320, [kinds[i % 4] for i in range(64)])).cuda()
pcm[17] = pcm[0]; pcm[63] = pcm[0]
eng.encode(eng.logmel(pcm)); eng.cross_kv(64)
prompt = np.tile(np.array(PROMPT, dtype=np.int32), (64, 1))
out = eng.generate_greedy(prompt, max_new_tokens=new, timestamps=True, want_alignment=True)
s = out["sequences"]
def first_diff(a, b):
    d = np.nonzero(a != b)[0]
    return int(d[0]) if len(d) else None
print(f"enc={dims['enc_layers']} mode={os.environ.get('TW_SK_CG_MODE','default')} cus={os.environ.get('THEWHISPER_DECODE_CUS','160')} layers={layers} T={T} graph={graph} new={new}: "
      f"first difference 17 vs 0: {first_diff(s[17], s[0])}, 63 vs 0: {first_diff(s[63], s[0])}, 17 vs 63: {first_diff(s[17], s[63])}", flush=True)
eng.close()
