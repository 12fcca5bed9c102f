"""Engine (fp8 context, micro, 1 decoder layer) against the two restatements of the cross-query order.  Run per setting:
TW_FUSE_CQ=0/1 python tools/dbg/dbg_fuse_fp8.py"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import whisper_oracle as wo
from tests.test_gpu_parity import dims_variant, make_engine, clips, PROMPT, rel_l2

dims = dims_variant("micro", enc_layers=1, dec_layers=1)
w = wo.make_weights(dims, 2)
B, T = 3, 100
for dtype in ("fp8", "bf16"):
    eng = make_engine(dims, w, T=T, max_batch=B, dtype=dtype)
    mel = wo.log_mel(clips(T * 320, B), dims.n_mels)
    enc = eng.encode(torch.from_numpy(mel).cuda(), return_hidden=True).cpu().numpy()
    eng.cross_kv(B); eng.decoder_reset(B)
    got = eng.decode_step([PROMPT[0]] * B).cpu().numpy()
    np.save(f"gpurun_out/dbg_{dtype}_fuse{os.environ.get('TW_FUSE_CQ','1')}.npy", got)
    if dtype == "fp8":
        ids = np.tile(np.array(PROMPT[:1]), (B, 1))
        for ahead in (True, False):
            oq = wo.OracleWhisperMXFP8(dims, w, T=T, cross_q_ahead=ahead)
            ref = oq.decode(ids, oq.new_cache(enc))[0][:, 0]
            print(dtype, "fuse", os.environ.get("TW_FUSE_CQ", "1"), "oracle ahead", ahead, rel_l2(got, ref))
    eng.close()
