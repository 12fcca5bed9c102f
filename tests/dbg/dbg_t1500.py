import sys, os, numpy as np, torch
sys.path.insert(0, os.getcwd())
from oracle import whisper_oracle as wo
from tests.util import PROMPT, clips, make_engine
for preset, T, B in [("micro80", 1500, 2), ("micro", 750, 2)]:
    dims = wo.PRESETS[preset]; w = wo.make_weights(dims, 0)
    heads = [(dims.dec_layers - 1, 0), (dims.dec_layers - 1, 1)]
    eng = make_engine(dims, w, T=T, max_batch=B, dtype="f32", heads=heads)
    pcm = clips(T * 320, B)
    eng.encode(eng.logmel(torch.from_numpy(pcm).cuda(), out_dtype=torch.float32)); eng.cross_kv(B)
    prompt = np.tile(np.array(PROMPT, dtype=np.int32), (B, 1))
    out = eng.generate_greedy(prompt, max_new_tokens=24, timestamps=True, want_alignment=True)
    om = wo.OracleWhisper(dims, w, T=T)
    opt = wo.GreedyOptions(max_new_tokens=24, timestamps=True, alignment_heads=heads)
    ref = wo.greedy_generate(om, om.encode(wo.log_mel(pcm, dims.n_mels)), prompt, opt)
    print(preset, T, "ids equal", np.array_equal(out["sequences"], ref["sequences"]))
    L = out["length"]
    al = eng.get_alignment(B, L - 1)
    print(" align maxabs", np.abs(al - ref["cross"]).max(), "rowsum", np.abs(al.sum(-1)-1).max())
    for nf in ([2*T]*B, [2*T-100, 2*T-1220]):
        ts = eng.token_timestamps(B, 3, L, nf); rts = wo.token_timestamps(ref["cross"], 3, nf)
        print(" nf", nf, "ts maxdiff", np.abs(ts-rts).max())
        if np.abs(ts-rts).max() > 0.03:
            print(ts[0][:14]); print(rts[0][:14]); print(ts[1][:14]); print(rts[1][:14])
    eng.close()
