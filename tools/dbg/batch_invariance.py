#!/usr/bin/env python3
"""Diagnostic (MI355X): is a clip's result independent of how many other clips share its pass?  Encoder states and teacher-forced
logits of slot 0 with 1, 2, 16, 17 and 64 streams, at the real width.
    python tools/dbg/batch_invariance.py <dtype> <T> [enc_layers] [dec_layers]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from oracle import whisper_oracle as wo
from tests.util import PROMPT, clips, dims_variant, make_engine

dtype, T = sys.argv[1], int(sys.argv[2])
el = int(sys.argv[3]) if len(sys.argv) > 3 else 2
dl = int(sys.argv[4]) if len(sys.argv) > 4 else 2
dims = dims_variant("large-v3", enc_layers=el, dec_layers=dl)
w = wo.make_weights(dims, 2)
eng = make_engine(dims, w, T=T, max_batch=64, dtype=dtype, heads=[(dl - 1, 0)], use_graph=False)
pcm = clips(T * 320, 64)
mel = eng.logmel(torch.from_numpy(pcm).cuda())
ids = np.concatenate([np.tile(np.array(PROMPT), (64, 1)), np.random.default_rng(3).integers(0, 50000, size=(64, 3))], axis=1)
ref_e = ref_l = None
for B in (1, 2, 16, 17, 64):
    e = eng.encode(mel[:B], return_hidden=True)[0].float().cpu().numpy()
    eng.cross_kv(B); eng.decoder_reset(B)
    lg = np.stack([eng.decode_step(ids[:B, s].tolist()).cpu().numpy()[0] for s in range(ids.shape[1])])
    if ref_e is None:
        ref_e, ref_l = e, lg
    print(f"{dtype} T={T} B={B}: encoder slot 0 differing elements {int((e != ref_e).sum())} of {e.size} (max |d| {np.abs(e - ref_e).max():.3e}), "
          f"logits differing {int((lg != ref_l).sum())} of {lg.size} (max |d| {np.abs(lg - ref_l).max():.3e})", flush=True)
eng.close()
