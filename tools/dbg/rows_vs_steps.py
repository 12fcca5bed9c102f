#!/usr/bin/env python3
"""Diagnostic (MI355X): a forced-prefix call (rows mode: launches of up to 64 rows) against the step-by-step call of the same context -
ids and alignment rows must be bit-identical (round 6).  Prints WHERE they differ.
    python tools/dbg/rows_vs_steps.py <dtype> <B> <n_forced> [repeats]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from oracle import whisper_oracle as wo
from tests.util import PROMPT, clips, dims_variant, make_engine

dtype, B, nfo = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
rep = int(sys.argv[4]) if len(sys.argv) > 4 else 3
dims = dims_variant("large-v3", enc_layers=1, dec_layers=int(os.environ.get("DEC_LAYERS", "2")))
w = wo.make_weights(dims, 2)
heads = [(dims.dec_layers - 1, 0), (dims.dec_layers - 1, 3)]
T = 100
eng = make_engine(dims, w, T=T, max_batch=64, dtype=dtype, heads=heads, use_graph=False)
pcm = clips(T * 320, 64)
mel = eng.logmel(torch.from_numpy(pcm).cuda())
eng.encode(mel[:B]); eng.cross_kv(B)
prompt = np.tile(np.array(PROMPT, dtype=np.int32), (B, 1))
kw = dict(max_new_tokens=nfo + 10, min_new_tokens=nfo + 10, timestamps=True, want_alignment=True)
full = eng.generate_greedy(prompt, **kw)
L = full["length"]
al0 = eng.get_alignment(B, L - 1)
for r in range(rep):
    again = eng.generate_greedy(prompt, **kw)
    al1 = eng.get_alignment(B, L - 1)
    print(f"step-mode repeat {r}: ids equal {np.array_equal(again['sequences'], full['sequences'])}, alignment equal {np.array_equal(al1, al0)}")
forced = full["sequences"][:, : 3 + nfo].astype(np.int32)
for r in range(rep):
    out = eng.generate_greedy(forced, n_forced=nfo, **kw)
    al = eng.get_alignment(B, L - 1)
    d = np.abs(al - al0)
    bad = np.argwhere(d.max(axis=-1) > 0)
    print(f"rows-mode repeat {r}: ids equal {np.array_equal(out['sequences'], full['sequences'])}, alignment rows differing: {len(bad)} of {B * 2 * (L - 1)}, max |d| {d.max():.3e}")
    if len(bad):
        pos = sorted(set(int(x[2]) for x in bad)); strm = sorted(set(int(x[0]) for x in bad)); hd = sorted(set(int(x[1]) for x in bad))
        print(f"   positions {pos[:40]}{'...' if len(pos) > 40 else ''} streams {strm} heads {hd}")
dump = os.environ.get("TW_DUMP")
if dump:   # internal buffers after ONE rows-mode launch of the first layer (run with DEC_LAYERS=1 and a small n_forced)
    import ctypes as C
    eng.generate_greedy(forced, n_forced=nfo, max_new_tokens=nfo + 1, min_new_tokens=nfo + 1, timestamps=True, want_alignment=True)
    out = {}
    d = dims.d_model
    for name, n in (("dq", 64 * d * 2), ("du", 64 * d * 4), ("dstats", 64 * (d // 4) * 2 * 4), ("datt", 64 * d * 2), ("dx0", 64 * d * 2), ("dx1", 64 * d * 2),
                    ("dh", 64 * dims.ffn * 2), ("self_k", 64 * 448 * d * 2), ("self_v", 64 * 448 * d * 2)):
        buf = np.zeros(n, dtype=np.uint8)
        rc = eng.lib.tw_dbg_copy(eng.ctx, name.encode(), buf.ctypes.data_as(C.c_void_p), C.c_size_t(n))
        assert rc == 0, name
        out[name] = buf
    np.savez_compressed(dump, **out)
eng.close()
