import faulthandler, sys, os
faulthandler.enable(all_threads=True)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from oracle import whisper_oracle as wo
from tests.util import PROMPT, clips, make_engine
def P(*a): print(*a, flush=True)
dims = wo.PRESETS["micro"]; w = wo.make_weights(dims, 0)
heads = [(dims.dec_layers - 1, 0), (dims.dec_layers - 1, 1)]
engs = [make_engine(dims, w, T=100, max_batch=3, dtype="f32", heads=heads, use_graph=True) for _ in range(2)]
P("engines ok")
from thewhisper_amd.overlap import EncoderOverlap, _hiplib
hip = _hiplib(); P("hiplib ok")
ev = hip.event_create(0); P("event ok", ev)
ov = EncoderOverlap(engs, encoder_cus=32); P("overlap ctor ok", ov.s_dec, ov.s_enc)
pc = torch.from_numpy(np.ascontiguousarray(clips(100 * 320, 3))).cuda()
e = engs[0]
e.raw_stream = ov.s_enc
ext = torch.cuda.ExternalStream(e.raw_stream, device=e.device); P("ext stream ok")
ext.wait_stream(torch.cuda.current_stream(e.device)); P("wait_stream ok")
mel = e.logmel(pc, out_dtype=torch.float32); P("logmel ok")
e.encode(mel); e.cross_kv(3); P("encode ok")
hip.event_record(ev, ov.s_enc); hip.stream_wait_event(ov.s_dec, ev); P("event record/wait ok")
e.raw_stream = ov.s_dec
prompt = np.tile(np.array(PROMPT, dtype=np.int32), (3, 1))
out = e.generate_greedy(prompt, max_new_tokens=8, timestamps=True, want_alignment=True); P("generate ok", out["length"])
hip.stream_synchronize(ov.s_dec); P("sync ok")
e.raw_stream = None
ov.close(); P("close ok")
for x in engs: x.close()
P("done")
