#!/usr/bin/env python3
"""Round 5 diagnostic: every stream of a 64- (32-) stream launch carries the SAME audio and the SAME tokens, so every row of every
projection launch must be bit-identical to row 0.  Prints which streams deviate (by group of 16 / lane) per TW_SK_CG_MODE."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import numpy as np, torch
from oracle import whisper_oracle as wo
from tests.util import PROMPT, clips, dims_variant, make_engine

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
layers = int(sys.argv[2]) if len(sys.argv) > 2 else 1
dims = dims_variant("large-v3", enc_layers=1, dec_layers=layers)
w = wo.make_weights(dims, 2)
T = 100
eng = make_engine(dims, w, T=T, max_batch=B, dtype="bf16")
one = clips(T * 320, 1)
mel = wo.log_mel(np.repeat(one, B, axis=0), dims.n_mels)
eng.encode(torch.from_numpy(mel).cuda()); eng.cross_kv(B); eng.decoder_reset(B)
ids = list(PROMPT) + [100, 2000, 31000]
for s, t in enumerate(ids):
    lg = eng.decode_step([t] * B).cpu().numpy()
    dev = np.abs(lg - lg[0:1]).max(axis=1)
    bad = np.nonzero(dev > 0)[0]
    print(f"mode={os.environ.get('TW_SK_CG_MODE','default')} B={B} step {s}: streams differing from stream 0: {bad.tolist()}  max dev {dev.max():.3e}", flush=True)
eng.close()
