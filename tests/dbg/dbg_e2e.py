import sys, os, json, numpy as np, torch
sys.path.insert(0, os.getcwd())
torch.set_grad_enabled(False)
from oracle import whisper_oracle as wo
from tests.test_pipeline_glue import build_amd_pipeline, normalise
from tests.oracle_engine import oracle_engine_factory
g = json.load(open("tests/golden/pipeline_golden.json"))["micro80_c30"]
audio = wo.synth_audio(16000 * g["seconds"], g["seed"], g["kind"])
gk = {"num_beams": 1, "do_sample": False, "use_cache": True, "language": "en", "max_new_tokens": g["max_new_tokens"]}
logs = {}
for tag, dev, fac in [("gpu", "cuda", None), ("cpu", "cpu", oracle_engine_factory)]:
    pipe = build_amd_pipeline(g["preset"], g["chunk_s"], g["batch_size"], device=dev, engine_factory=fac)
    m = pipe.model
    orig = m._extract_token_timestamps
    rec = []
    def spy(generate_outputs, alignment_heads, time_precision=0.02, num_frames=None, num_input_ids=None, _o=orig, _r=rec):
        out = _o(generate_outputs, alignment_heads, time_precision, num_frames, num_input_ids)
        nf = num_frames.tolist() if hasattr(num_frames, "tolist") else num_frames
        _r.append((tuple(generate_outputs["sequences"].shape), nf, num_input_ids, generate_outputs["sequences"].cpu().numpy().copy(), out.cpu().numpy().copy()))
        return out
    m._extract_token_timestamps = spy
    out = normalise(pipe(audio.copy(), generate_kwargs=dict(gk), chunk_length_s=g["chunk_s"] - 1, return_timestamps="word"))
    logs[tag] = (rec, out)
ga, ca = logs["gpu"][0], logs["cpu"][0]
print("calls", len(ga), len(ca))
for i, (a, b) in enumerate(zip(ga, ca)):
    print(i, "shape", a[0], b[0], "nf", a[1], b[1], "n_in", a[2], b[2], "ids equal", np.array_equal(a[3], b[3]), "ts maxdiff", np.abs(a[4] - b[4]).max() if a[4].shape == b[4].shape else "shape")
    if a[4].shape == b[4].shape and np.abs(a[4] - b[4]).max() > 0.03:
        bi = int(np.argmax(np.abs(a[4] - b[4]).max(axis=1)))
        print("  gpu", a[4][bi][:20]); print("  cpu", b[4][bi][:20]); print("  ids", a[3][bi][:20])
print("final equal:", logs["gpu"][1] == logs["cpu"][1], "cpu==golden:", logs["cpu"][1] == g["outputs"]["word"])
