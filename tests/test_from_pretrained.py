"""The constructor form the reference's examples actually use: a checkpoint NAME / PATH string.

    R:examples/run_nvidia_asr.py:22-35     ASRPipeline("TheStageAI/thewhisper-large-v3-turbo", chunk_length_s=10, model_size="S",
                                                       batch_size=..., device="cuda")
    R:thestage_speechkit/nvidia/asr_pipeline.py:47-70   model / feature extractor / tokenizer `from_pretrained(model_name, ...)`

There is no hub here, so the checkpoint is a local directory written by `save_pretrained` (micro model, synthetic tokenizer,
feature extractor saved with the upstream default chunk_length = 30, generation_config with alignment_heads), in bf16 AND
fp16 safetensors.  Covered: tied `proj_out` (absent from the safetensors file), the `revision` kwarg, the feature extractor's
`chunk_length` override (30 on disk, 10 asked), `model_size="S"`, and `python -m thewhisper_amd.gateway --model <dir>` through
`build_host` + `create_app`.  The result must equal what the INSTANCE-constructed pipeline (the form every other test uses)
returns for the same weights.  CPU: oracle-backed engine through the class-level factory seam; `-m gpu`: the HIP engine.
"""
import io
import json
import os
import wave

import numpy as np
import pytest
import torch

from oracle import hf_reference as hr
from oracle import whisper_oracle as wo
from tests.oracle_engine import oracle_engine_factory

torch.set_grad_enabled(False)
GK = {"num_beams": 1, "do_sample": False, "use_cache": True, "language": "en", "max_new_tokens": 24}


def save_checkpoint(tmp, dtype):
    dims = wo.PRESETS["micro"]
    model = hr.build_hf_model(dims, wo.make_weights(dims, 0)).to(dtype)
    model.generation_config._from_model_config = False   # a hand-filled generation config, as the published checkpoints carry
    d = str(tmp)
    model.save_pretrained(d)
    hr.build_tokenizer(dims).save_pretrained(d)
    hr.build_feature_extractor(dims, 30).save_pretrained(d)    # upstream preprocessor_config.json says chunk_length 30
    return d, dims


def normalise(out):
    return json.loads(json.dumps(out))


def instance_pipeline(d, dims, chunk_s, batch, device, dtype, engine_factory):
    """The other constructor form: an HF model instance + explicit feature extractor / tokenizer (same rounded weights)."""
    from transformers import WhisperForConditionalGeneration

    from thewhisper_amd import ASRPipeline

    model = WhisperForConditionalGeneration.from_pretrained(d, dtype=dtype)
    kw = {} if engine_factory is None else {"engine_factory": engine_factory}
    return ASRPipeline(model, feature_extractor=hr.build_feature_extractor(dims, chunk_s), tokenizer=hr.build_tokenizer(dims),
                       chunk_length_s=chunk_s, device=device, torch_dtype=dtype, batch_size=batch, **kw)


def run_both(tmp_path, dtype, device, engine_factory):
    from safetensors import safe_open

    from thewhisper_amd import ASRPipeline
    from thewhisper_amd.model import AMDWhisperForConditionalGeneration

    d, dims = save_checkpoint(tmp_path, dtype)
    with safe_open(os.path.join(d, "model.safetensors"), "pt") as f:
        keys = set(f.keys())
        assert "proj_out.weight" not in keys and "model.decoder.embed_tokens.weight" in keys      # tied: stored once
        assert f.get_tensor("model.decoder.embed_tokens.weight").dtype == dtype
    kw = {} if engine_factory is None else {"engine_factory": engine_factory}
    # exactly the reference example's call (R:examples/run_nvidia_asr.py:22-35), plus the `revision` kwarg its constructor pops
    pipe = ASRPipeline(d, chunk_length_s=10, model_size="S", batch_size=4, device=device, revision="main", **kw)
    assert isinstance(pipe.model, AMDWhisperForConditionalGeneration)
    assert pipe.feature_extractor.chunk_length == 10                       # override of the 30 on disk
    assert pipe.model.config.max_source_positions == 500                  # A0 happened for T = 500
    assert [list(h) for h in pipe.model.generation_config.alignment_heads] == [list(h) for h in hr.default_alignment_heads(dims)]
    ref = instance_pipeline(d, dims, 10, 4, device, dtype, engine_factory)
    audio = wo.synth_audio(16000 * 25, 3, "speechlike")
    for rt in (False, True, "word"):
        got = pipe(audio.copy(), generate_kwargs=dict(GK), chunk_length_s=9, return_timestamps=rt)
        want = ref(audio.copy(), generate_kwargs=dict(GK), chunk_length_s=9, return_timestamps=rt)
        assert normalise(got) == normalise(want), f"return_timestamps={rt}"
        assert len(got["text"]) > 0
    return d, dims, pipe


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16], ids=["bf16", "fp16"])
def test_string_constructor_equals_instance_constructor_cpu(tmp_path, dtype):
    run_both(tmp_path, dtype, "cpu", oracle_engine_factory)


def test_wrong_chunk_length_and_model_size_raise_like_the_reference(tmp_path):
    from thewhisper_amd import ASRPipeline

    d, _ = save_checkpoint(tmp_path, torch.bfloat16)
    with pytest.raises(ValueError, match="Invalid model_size"):           # R:thestage_speechkit/nvidia/asr_pipeline.py:44-45
        ASRPipeline(d, model_size="XXL", device="cpu", engine_factory=oracle_engine_factory)
    with pytest.raises(OSError):                                           # no such checkpoint: HF's own error, not a fallback
        ASRPipeline(os.path.join(d, "missing"), device="cpu", engine_factory=oracle_engine_factory)


def _wav(audio, sr=16000):
    pcm = (np.clip(audio.astype(np.float32), -1.0, 1.0) * 32767.0).astype(np.int16)
    buf = io.BytesIO()
    with wave.open(buf, "wb") as wf:
        wf.setnchannels(1); wf.setsampwidth(2); wf.setframerate(sr); wf.writeframes(pcm.tobytes())
    return buf.getvalue()


def gateway_from_checkpoint(d, monkeypatch, device, engine_factory):
    """`python -m thewhisper_amd.gateway --model <dir> ...` up to the point where uvicorn would take over."""
    pytest.importorskip("fastapi")
    from fastapi.testclient import TestClient

    from thewhisper_amd.gateway import build_host, create_app, decode_wav
    from thewhisper_amd.model import AMDWhisperForConditionalGeneration

    monkeypatch.setenv("THEWHISPER_DEVICE", device)     # the backend says "cuda" like the reference's (streaming_pipeline.py:365)
    if engine_factory is not None:
        monkeypatch.setattr(AMDWhisperForConditionalGeneration, "_engine_factory", staticmethod(engine_factory))
    host, args = build_host(["--model", d, "--chunk-length-s", "10", "--max-batch", "4", "--language", "en"])
    try:
        client = TestClient(create_app(host, model_name=args.model, lang_id=args.language))
        audio = wo.synth_audio(16000 * 6, 11, "speechlike")
        r = client.post("/transcribe", files={"file": ("chunk.wav", _wav(audio), "audio/wav")}, headers={"X-Lang-Id": "en"})
        assert r.status_code == 200, r.text
        data = r.json()
        got = [{"text": c["text"], "start": c["timestamp"][0], "end": c["timestamp"][1]} for c in data["metadata"]["chunks"]]
        want = host.base_backend.transcribe(decode_wav(_wav(audio))[0], 0.0, 16000)
        assert normalise(got) == normalise(want) and len(want) > 0
        assert client.get("/health").json()["passes"] >= 1
    finally:
        host.hub.close()
    return host


def test_gateway_command_line_builds_from_a_checkpoint_path_cpu(tmp_path, monkeypatch):
    d, _ = save_checkpoint(tmp_path, torch.bfloat16)
    host = gateway_from_checkpoint(d, monkeypatch, "cpu", oracle_engine_factory)
    assert host.base_backend.asr_pipeline.feature_extractor.chunk_length == 10


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16], ids=["bf16", "fp16"])
def test_string_constructor_on_the_mi355x(tmp_path, dtype):
    if not torch.cuda.is_available():
        pytest.skip("needs the MI355X")
    _, _, pipe = run_both(tmp_path, dtype, "cuda", None)
    eng = pipe.model.engine
    assert type(eng).__name__ == "WhisperEngine" and eng.T == 500 and eng.max_batch == 4
    eng.close()


@pytest.mark.gpu
def test_gateway_command_line_on_the_mi355x(tmp_path, monkeypatch):
    if not torch.cuda.is_available():
        pytest.skip("needs the MI355X")
    d, _ = save_checkpoint(tmp_path, torch.bfloat16)
    gateway_from_checkpoint(d, monkeypatch, "cuda", None)
