"""The `amd` platform as shipped: integration/apply.py patches a COPY of the reference checkout (amd/__init__.py + the
four-line branch of the platform switch, R:thestage_speechkit/streaming/streaming_pipeline.py:358-367), and the reference's
own StreamingPipeline(platform="amd") then runs end to end.  CPU: the numpy oracle is injected as the engine; the words
must equal the golden stream the reference produced with its own backend."""
import importlib
import json
import os
import sys
import types

import numpy as np
import pytest
import torch

from oracle import hf_reference as hr
from oracle import ref_bundle
from oracle import whisper_oracle as wo
from tests.oracle_engine import oracle_engine_factory

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# /root/reference in the build container; on the GPU box the package __graft_entry__.build() packed (oracle/ref_bundle.py)
REF = ref_bundle.reference_dir() or "/root/reference"
GOLD = os.path.join(ROOT, "tests", "golden", "pipeline_golden.json")


def _purge():
    for k in [k for k in sys.modules if k == "thestage_speechkit" or k.startswith("thestage_speechkit.")]:
        del sys.modules[k]


@pytest.fixture()
def patched_reference(tmp_path, monkeypatch):
    if not os.path.isdir(os.path.join(REF, "thestage_speechkit")):
        pytest.skip("neither the reference checkout nor its oracle/_ref bundle is present")
    sys.path.insert(0, os.path.join(ROOT, "integration"))
    try:
        import apply as tw_apply
    finally:
        sys.path.pop(0)
    out = tw_apply.apply(REF, str(tmp_path / "ref"))
    assert tw_apply.apply(REF, out) == out                       # idempotent
    import transformers  # noqa: F401  (before the audio-I/O stubs)

    for name in ("sounddevice", "librosa"):
        if name not in sys.modules:
            m = types.ModuleType(name)
            m.InputStream = type("InputStream", (), {})
            m.load = m.resample = lambda *a, **k: (_ for _ in ()).throw(RuntimeError("stub"))
            monkeypatch.setitem(sys.modules, name, m)
    _purge()
    monkeypatch.syspath_prepend(out)
    from transformers.models.whisper import tokenization_whisper as _tw

    saved = _tw._find_longest_common_sequence     # importing the reference package installs ITS chunk-merge patch
    yield out
    _tw._find_longest_common_sequence = saved
    _purge()
    import thewhisper_amd.lcs_patch as lp   # the reference package is gone again: our equivalent of its patch takes over

    lp.install()


def test_patch_touches_only_the_platform_switch(patched_reference):
    a = open(os.path.join(REF, "thestage_speechkit/streaming/streaming_pipeline.py")).read().splitlines()
    b = open(os.path.join(patched_reference, "thestage_speechkit/streaming/streaming_pipeline.py")).read().splitlines()
    added = [l for l in b if l not in a]
    assert len(b) - len(a) == 4 and [l.strip() for l in added if l.strip()] == ['elif platform == "amd":', "from ..amd import ASRPipeline"]
    assert os.path.isfile(os.path.join(patched_reference, "thestage_speechkit/amd/__init__.py"))


def test_streaming_pipeline_platform_amd_reproduces_the_reference_stream(patched_reference, monkeypatch):
    from thewhisper_amd.model import AMDWhisperForConditionalGeneration

    monkeypatch.setenv("THEWHISPER_DEVICE", "cpu")
    monkeypatch.setattr(AMDWhisperForConditionalGeneration, "_engine_factory", staticmethod(oracle_engine_factory))
    sp = importlib.import_module("thestage_speechkit.streaming.streaming_pipeline")
    assert "thestage_speechkit/streaming" in sp.__file__ and patched_reference in sp.__file__
    g = json.load(open(GOLD))["streaming_micro_c10"]
    dims = wo.PRESETS["micro"]
    model = hr.build_hf_model(dims, wo.make_weights(dims, 0))
    stream = sp.StreamingPipeline(model=model, platform="amd", chunk_length_s=10, min_process_chunk_s=0.5, use_vad=False,
                                  torch_dtype=torch.float32, language="en",
                                  feature_extractor=hr.build_feature_extractor(dims, 10), tokenizer=hr.build_tokenizer(dims))
    import thestage_speechkit.amd as amd_pkg
    from thewhisper_amd import ASRPipeline

    assert amd_pkg.ASRPipeline is ASRPipeline and isinstance(stream.backend.asr_pipeline, ASRPipeline)
    audio = wo.synth_audio(16000 * g["seconds"], g["seed"], g["kind"])
    committed, last = [], []
    for i in range(0, len(audio), g["step_samples"]):
        c, u = stream(audio[i : i + g["step_samples"]])
        committed += c
        last = u
    assert json.loads(json.dumps(committed)) == g["committed"]
    assert json.loads(json.dumps(last)) == g["uncommitted"]
    with pytest.raises(ValueError, match="Invalid platform"):
        sp.LocalWhisperBackend(model, platform="tpu")


@pytest.mark.gpu
def test_reference_scheduler_drives_the_mi355x(patched_reference):
    """BASELINE config 3's defining call path ON THE HARDWARE: the reference's own `StreamingPipeline(platform="amd")`
    (R:thestage_speechkit/streaming/streaming_pipeline.py:358-367 + the four-line patch, :443-531, :740-822), fed by the reference's
    own `ArrayStream(step_size_s=0.05, real_time=False)` (R:thestage_speechkit/streaming/streams.py:16-81), builds
    `thewhisper_amd.ASRPipeline` on the real engine (strict-f32 context on cuda:0, no stand-in) and must reproduce the committed /
    uncommitted words the reference produced with its own backend on CPU (tests/golden: streaming_micro_c10)."""
    assert torch.cuda.is_available()
    sp = importlib.import_module("thestage_speechkit.streaming.streaming_pipeline")
    streams = importlib.import_module("thestage_speechkit.streaming.streams")
    assert patched_reference in sp.__file__
    g = json.load(open(GOLD))["streaming_micro_c10"]
    dims = wo.PRESETS["micro"]
    model = hr.build_hf_model(dims, wo.make_weights(dims, 0))
    stream = sp.StreamingPipeline(model=model, platform="amd", chunk_length_s=10, min_process_chunk_s=0.5, use_vad=False,
                                  torch_dtype=torch.float32, language="en",
                                  feature_extractor=hr.build_feature_extractor(dims, 10), tokenizer=hr.build_tokenizer(dims))
    from thewhisper_amd import ASRPipeline
    from thewhisper_amd.engine import WhisperEngine

    pipe = stream.backend.asr_pipeline
    assert isinstance(pipe, ASRPipeline) and isinstance(pipe.model.engine, WhisperEngine) and stream.backend.device == "cuda"
    calls = []
    inner = stream.backend.transcribe

    def spy(audio, buffer_start_time, sample_rate):
        calls.append((len(audio), float(buffer_start_time)))
        return inner(audio, buffer_start_time, sample_rate)

    stream.backend.transcribe = spy
    audio = wo.synth_audio(16000 * g["seconds"], g["seed"], g["kind"])
    src = streams.ArrayStream(audio, step_size_s=g["step_samples"] / 16000, sample_rate=16000, real_time=False)
    committed, last = [], []
    while True:
        chunk = src.next_chunk()
        if chunk is None:
            break
        c, u = stream(chunk)
        committed += c
        last = u

    def same(a, b):
        assert [w["text"] for w in a] == [w["text"] for w in b]
        for x, y in zip(a, b):     # word timestamps: DTW on float32 statistics, within one 0.02 s frame (the GPU suite's stated bound)
            assert abs(x["start"] - y["start"]) <= 0.0201 and abs(x["end"] - y["end"]) <= 0.0201

    same(json.loads(json.dumps(committed)), g["committed"])
    same(json.loads(json.dumps(last)), g["uncommitted"])
    # ... and the scheduler asked for exactly the buffers it asked the reference backend for
    assert [(n, round(t0, 6)) for n, t0 in calls] == [(c["n"], round(c["t0"], 6)) for c in g["calls"]]
