"""The oracle against the committed golden vectors that oracle/make_golden.py produced by running the REFERENCE's own
``thestage_speechkit.nvidia.ASRPipeline`` (HF branch) on CPU.  Weights are regenerated from the seed."""
import json
import os

import numpy as np
import pytest

from oracle import whisper_oracle as wo

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def load_case(name):
    with open(os.path.join(GOLD, "pipeline_golden.json")) as f:
        meta = json.load(f)[name]
    return meta, np.load(os.path.join(GOLD, name + ".npz"))


@pytest.mark.parametrize("name", ["micro_c10", "micro80_c30", "micro_c10_noise", "tiny_en_c30"])
def test_oracle_reproduces_reference_vectors(name):
    meta, z = load_case(name)
    dims = wo.PRESETS[meta["preset"]]
    w = wo.make_weights(dims, meta["weight_seed"], **meta.get("weight_kw", {}))
    T = 50 * meta["chunk_s"]
    audio = wo.synth_audio(16000 * meta["seconds"], meta["seed"], meta["kind"])
    clip = audio[: meta["chunk_s"] * 16000]
    mel = wo.log_mel(clip, dims.n_mels, meta["chunk_s"] * 16000)
    assert np.abs(mel[0, ::16, ::25] - z["mel_rows"]).max() < 5e-5
    om = wo.OracleWhisper(dims, w, T=T)
    enc = om.encode(mel)
    assert np.abs(enc[0, ::50, ::8] - z["enc_rows"]).max() < 5e-5
    lg, _ = om.decode(z["teacher_ids"].astype(np.int64), om.new_cache(enc))
    top_idx = np.argsort(-lg[0], axis=-1)[:, :8]
    assert np.array_equal(top_idx, z["logits_top_idx"])
    assert np.abs(np.take_along_axis(lg[0], top_idx, -1) - z["logits_top"]).max() < 5e-5


def test_golden_metadata():
    with open(os.path.join(GOLD, "pipeline_golden.json")) as f:
        meta = json.load(f)["_meta"]
    assert meta["generator"] == "oracle/make_golden.py" and meta["transformers"].startswith("5.")
