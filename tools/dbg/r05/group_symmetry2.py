#!/usr/bin/env python3
"""Round 5 diagnostic 2: 64 streams with DIFFERENT audio except that streams 17 and 63 repeat stream 0; teacher-forced logits.
(a) streams 17 / 63 against stream 0 (must be bit-identical), (b) every stream against the same stream decoded in a 16-stream
context (bf16-rounding-level differences are legitimate: fc2 splits K differently above 16 streams)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import numpy as np, torch
from oracle import whisper_oracle as wo
from tests.util import PROMPT, clips, dims_variant, make_engine

B = 64
layers = int(sys.argv[1]) if len(sys.argv) > 1 else 1
dims = dims_variant("large-v3", enc_layers=1, dec_layers=layers)
w = wo.make_weights(dims, 2)
T = 100
kinds = ["speechlike", "noise", "sine", "speechlike"]
pcm = clips(T * 320, [kinds[i % 4] for i in range(B)])
pcm[17] = pcm[0]; pcm[63] = pcm[0]
mel = wo.log_mel(pcm, dims.n_mels)
ids = np.concatenate([np.tile(np.array(PROMPT), (B, 1)), np.random.default_rng(3).integers(0, 50000, size=(B, 4))], axis=1)
ids[17] = ids[0]; ids[63] = ids[0]

def run(eng, sel):
    n = len(sel)
    eng.encode(torch.from_numpy(mel[sel]).cuda()); eng.cross_kv(n); eng.decoder_reset(n)
    return np.stack([eng.decode_step(ids[sel, s].tolist()).cpu().numpy() for s in range(ids.shape[1])], axis=1)

big = make_engine(dims, w, T=T, max_batch=B, dtype="bf16")
got = run(big, np.arange(B)); big.close()
small = make_engine(dims, w, T=T, max_batch=16, dtype="bf16")
ref = np.concatenate([run(small, np.arange(lo, lo + 16)) for lo in (0, 16, 32, 48)]); small.close()
m = os.environ.get("TW_SK_CG_MODE", "default")
for s in range(ids.shape[1]):
    d17, d63 = np.abs(got[17, s] - got[0, s]).max(), np.abs(got[63, s] - got[0, s]).max()
    rel = np.linalg.norm(got[:, s] - ref[:, s], axis=1) / np.linalg.norm(ref[:, s], axis=1)
    worst = np.argsort(-rel)[:6]
    print(f"mode={m} layers={layers} step {s}: |17-0|={d17:.3e} |63-0|={d63:.3e}  rel-L2 vs 16-stream context: max {rel.max():.3e} median {np.median(rel):.3e} worst streams {worst.tolist()}"
          f" by group {[round(float(rel[g*16:(g+1)*16].max()),5) for g in range(4)]}", flush=True)
