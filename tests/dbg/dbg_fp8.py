import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import whisper_oracle as wo
from util import make_engine, dims_variant, clips, rel_l2, PROMPT
for preset, T, B in [("micro", 100, 3), ("large-v3", 500, 2)]:
    for layers in (0, 1, 2):
        dims = dims_variant(preset, enc_layers=1, dec_layers=layers)
        w = wo.make_weights(dims, 2)
        eng = make_engine(dims, w, T=T, max_batch=B, dtype="fp8")
        mel = wo.log_mel(clips(T * 320, B), dims.n_mels)
        om = wo.OracleWhisper(dims, w, T=T); oq = wo.OracleWhisperMXFP8(dims, w, T=T)
        enc = eng.encode(torch.from_numpy(mel).cuda(), return_hidden=True).cpu().numpy()
        eng.cross_kv(B); eng.decoder_reset(B)
        ids = np.tile(np.array(PROMPT), (B, 1))
        c, cq = om.new_cache(enc), oq.new_cache(enc)
        for s in range(2):
            ref = om.decode(ids[:, s:s+1], c)[0][:, 0]; refq = oq.decode(ids[:, s:s+1], cq)[0][:, 0]
            got = eng.decode_step(ids[:, s].tolist()).cpu().numpy()
            print(f"{preset} layers={layers} step={s}: got-vs-emulated {rel_l2(got, refq):.4f}  got-vs-exact {rel_l2(got, ref):.4f}  emulated-vs-exact {rel_l2(refq, ref):.4f}", flush=True)
        eng.close()
