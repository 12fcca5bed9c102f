"""Repeats the EncoderOverlap-vs-sequential comparison many times (race screen): python tests/dbg/stress_overlap.py [N]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import whisper_oracle as wo
from util import make_engine, clips, PROMPT
from thewhisper_amd.overlap import EncoderOverlap

N = int(sys.argv[1]) if len(sys.argv) > 1 else 40
dims = wo.PRESETS["micro"]; w = wo.make_weights(dims, 0)
heads = [(dims.dec_layers - 1, 0), (dims.dec_layers - 1, 1)]
engs = [make_engine(dims, w, T=100, max_batch=3, dtype="f32", heads=heads, use_graph=True) for _ in range(2)]
batches = [torch.from_numpy(np.ascontiguousarray(clips(100 * 320, 3)[::s])).cuda() for s in (1, -1, 1, -1, 1, -1)]
prompt = np.tile(np.array(PROMPT, dtype=np.int32), (3, 1))
def enc(e, pc):
    mel = e.logmel(pc, out_dtype=torch.float32); e.encode(mel); e.cross_kv(3); return None if os.environ.get("DROP_MEL") else mel
def dec(e, pc, _):
    out = e.generate_greedy(prompt, max_new_tokens=16, timestamps=True, want_alignment=True)
    return out["sequences"].copy(), e.get_alignment(3, out["length"] - 1).copy()
seq = []
for b in batches:
    enc(engs[0], b); seq.append(dec(engs[0], b, None))
ov = EncoderOverlap(engs, encoder_cus=32)
bad = 0
for it in range(N):
    got = ov.run(batches, enc, dec)
    for a, b in zip(got, seq):
        if not (np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])):
            bad += 1
print(f"stress_overlap: {N} runs x {len(batches)} batches, mismatching batches: {bad}")
