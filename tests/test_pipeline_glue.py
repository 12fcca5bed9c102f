"""Host-side glue of the drop-in (ASRPipeline / AMDWhisperForConditionalGeneration / feature extractor / streaming
backend) on CPU, with the numpy oracle injected as the engine (tests/oracle_engine.py).  Outputs are compared with the
golden JSON the REFERENCE's own nvidia.ASRPipeline + StreamingPipeline produced (oracle/make_golden.py)."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import hf_reference as hr
from oracle import whisper_oracle as wo
from tests.oracle_engine import oracle_engine_factory

torch.set_grad_enabled(False)
GOLD = os.path.join(os.path.dirname(__file__), "golden", "pipeline_golden.json")


def golden():
    with open(GOLD) as f:
        return json.load(f)


def build_amd_pipeline(preset, chunk_s, batch_size, device="cpu", engine_factory=oracle_engine_factory,
                       dtype=torch.float32, weight_kw=None):
    from thewhisper_amd import ASRPipeline

    dims = wo.PRESETS[preset]
    w = wo.make_weights(dims, 0, **(weight_kw or {}))
    model = hr.build_hf_model(dims, w)
    kw = {} if engine_factory is None else {"engine_factory": engine_factory}
    return ASRPipeline(model, feature_extractor=hr.build_feature_extractor(dims, chunk_s), tokenizer=hr.build_tokenizer(dims),
                       chunk_length_s=chunk_s, device=device, torch_dtype=dtype, batch_size=batch_size, **kw)


def normalise(out):
    return json.loads(json.dumps(out))


@pytest.mark.parametrize("name", ["micro_c10", "micro_c10_noise", "micro80_c30", "tiny_en_c30"])
def test_offline_pipeline_matches_reference_golden(name):
    g = golden()[name]
    pipe = build_amd_pipeline(g["preset"], g["chunk_s"], g["batch_size"], weight_kw=g.get("weight_kw"))
    audio = wo.synth_audio(16000 * g["seconds"], g["seed"], g["kind"])
    gk = {"num_beams": 1, "do_sample": False, "use_cache": True, "language": "en", "max_new_tokens": g["max_new_tokens"]}
    for rt in (False, True, "word"):
        out = pipe(audio.copy(), generate_kwargs=dict(gk), chunk_length_s=g["chunk_s"] - 1, return_timestamps=rt)
        assert normalise(out) == g["outputs"][str(rt)], f"return_timestamps={rt}"
    eng = pipe.model.engine
    assert eng.calls["logmel"] > 0 and eng.calls["generate"] > 0 and eng.calls["dtw"] > 0  # all stages went through the engine


def test_streaming_backend_replays_reference_calls():
    """AMDWhisperBackend.transcribe == the reference's LocalWhisperBackend.transcribe on every rolling buffer the
    reference scheduler produced (ragged lengths, advancing buffer_start_time)."""
    from thewhisper_amd import AMDWhisperBackend

    g = golden()["streaming_micro_c10"]
    pipe = build_amd_pipeline("micro", 10, 1)
    backend = AMDWhisperBackend(None, chunk_length_s=10, asr_pipeline=pipe)
    audio = wo.synth_audio(16000 * g["seconds"], g["seed"], g["kind"])
    picks = g["calls"][::3] + g["calls"][-2:]
    for c in picks:
        buf = audio[c["offset"] : c["offset"] + c["n"]]
        got = backend.transcribe(buf, c["t0"], 16000)
        assert normalise(got) == c["result"], (c["n"], c["offset"], c["t0"])


def test_constructor_contract_mirrors_reference():
    from thewhisper_amd import ASRPipeline

    dims = wo.PRESETS["micro"]
    model = hr.build_hf_model(dims, wo.make_weights(dims, 0))
    with pytest.raises(ValueError, match="feature_extractor must be provided when passing a model instance"):
        ASRPipeline(model, tokenizer=hr.build_tokenizer(dims), device="cpu", engine_factory=oracle_engine_factory)
    with pytest.raises(ValueError, match="tokenizer must be provided when passing a model instance"):
        ASRPipeline(model, feature_extractor=hr.build_feature_extractor(dims, 30), device="cpu",
                    engine_factory=oracle_engine_factory)


def test_unsupported_generation_options_fail_loudly():
    pipe = build_amd_pipeline("micro", 10, 1)
    audio = wo.synth_audio(16000 * 5, 1, "speechlike")
    with pytest.raises(NotImplementedError, match="num_beams"):
        pipe(audio, generate_kwargs={"num_beams": 2, "language": "en", "max_new_tokens": 8})


def test_wrong_feature_length_raises_like_hf():
    pipe = build_amd_pipeline("micro", 10, 1)
    bad = torch.zeros(1, 128, 3000)
    with pytest.raises(ValueError, match="mel input features"):
        pipe.model.engine.encode(bad)


def test_no_engine_no_fallback():
    from thewhisper_amd.model import AMDWhisperForConditionalGeneration

    dims = wo.PRESETS["micro"]
    m = AMDWhisperForConditionalGeneration.from_hf(hr.build_hf_model(dims, wo.make_weights(dims, 0)))
    with pytest.raises(RuntimeError, match="no eager fallback"):
        m.generate(input_features=torch.zeros(1, 128, 3000), language="en")


def test_lcs_patch_is_the_reference_compare():
    from transformers.models.whisper import tokenization_whisper as tw

    import thewhisper_amd.lcs_patch as lp

    assert getattr(tw._find_longest_common_sequence, "_thewhisper_patched", False) or "thestage_speechkit" in __import__("sys").modules
    assert lp._ordered((1.0, None), (0.5, 0.7)) is True       # open-ended left token: in order
    assert lp._ordered((1.0, 1.2), (0.5, 0.7)) is False
    assert lp._ordered((0.2, 0.4), (0.5, 0.7)) is True


def test_shared_engine_and_skeleton_model_give_the_reference_result():
    """ASRPipeline(engine=...): a context loaded elsewhere under a weight-less skeleton model (thewhisper_amd/synthetic.py) -
    the construction bench.py's pipeline-level leg and multi-pipeline serving use - returns what the reference returned."""
    import dataclasses

    from thewhisper_amd import ASRPipeline, synthetic

    g = golden()["micro_c10"]
    dims = wo.PRESETS[g["preset"]]
    d = dataclasses.asdict(dims)
    heads = hr.default_alignment_heads(dims)
    eng = oracle_engine_factory(d, 500, g["batch_size"], "f32", heads, 0)
    eng.load_state_dict({k: torch.from_numpy(v) for k, v in wo.make_weights(dims, 0).items()})
    model = synthetic.skeleton_model(d, device="cpu", dtype=torch.float32, alignment_heads=heads)
    assert sum(p.numel() for p in model.parameters()) == 0
    pipe = ASRPipeline(model, feature_extractor=hr.build_feature_extractor(dims, 10), tokenizer=synthetic.build_tokenizer(dims.vocab),
                       chunk_length_s=10, device="cpu", torch_dtype=torch.float32, batch_size=g["batch_size"], engine=eng)
    audio = wo.synth_audio(16000 * g["seconds"], g["seed"], g["kind"])
    gk = {"num_beams": 1, "do_sample": False, "use_cache": True, "language": "en", "max_new_tokens": g["max_new_tokens"]}
    out = pipe(audio.copy(), generate_kwargs=dict(gk), chunk_length_s=9, return_timestamps="word")
    assert normalise(out) == g["outputs"]["word"]
    with pytest.raises(ValueError, match="encoder frames"):
        ASRPipeline(model, feature_extractor=hr.build_feature_extractor(dims, 30), tokenizer=synthetic.build_tokenizer(dims.vocab),
                    chunk_length_s=30, device="cpu", torch_dtype=torch.float32, engine=eng)


def test_special_id_cache_is_transparent():
    from thewhisper_amd.tokenizer_cache import cache_special_ids

    dims = wo.PRESETS["micro"]
    plain, tok = hr.build_tokenizer(dims), cache_special_ids(hr.build_tokenizer(dims))
    assert tok.all_special_ids == plain.all_special_ids and tok.all_special_ids is tok.all_special_ids   # cached object
    ids = [50258, 50259, 50360, 50365, 300, 301, 50400, 50257]
    assert tok._decode_with_timestamps(ids) == plain._decode_with_timestamps(ids)
    assert tok.decode(ids, skip_special_tokens=True) == plain.decode(ids, skip_special_tokens=True)
    tok.add_special_tokens({"additional_special_tokens": ["<|brandnew|>"]})      # table changes -> cache re-derived
    plain.add_special_tokens({"additional_special_tokens": ["<|brandnew|>"]})
    assert tok.all_special_ids == plain.all_special_ids
    assert cache_special_ids(tok) is tok


def test_short_decode_memo_is_transparent():
    """tokenizer_cache memoises decode() of <= 4 ids with default options (what HF's word splitting calls once per token):
    every such call returns what HF's decode returns, repeated calls hit the memo, other argument shapes bypass it, and a
    change of the special-token tables drops it."""
    from thewhisper_amd.tokenizer_cache import cache_special_ids

    dims = wo.PRESETS["micro"]
    plain, tok = hr.build_tokenizer(dims), cache_special_ids(hr.build_tokenizer(dims))
    rng = np.random.default_rng(0)
    pool = [50257, 50258, 50259, 50360, 50364, 50365, 50400, 51865, 0, 1, 195, 220, 255, 256, 300, 301, 4000, 50256]
    for _ in range(400):
        ids = [int(x) for x in rng.choice(pool, size=int(rng.integers(1, 6)))]
        for kw in ({}, {"decode_with_timestamps": True}, {"skip_special_tokens": True},
                   {"decode_with_timestamps": True, "skip_special_tokens": True}):
            assert tok.decode(ids, **kw) == plain.decode(ids, **kw), (ids, kw)
    memo = tok.__dict__["_tw_decode_memo"][1]
    assert 0 < len(memo) and all(len(k[0]) <= 4 for k in memo)
    n = len(memo)
    assert tok.decode([300], decode_with_timestamps=True) is tok.decode([300], decode_with_timestamps=True)   # memo hit
    assert len(memo) == n
    # bypass: tensors / numpy ids / other options go to HF's decode and are not stored
    assert tok.decode(np.array([300, 301])) == plain.decode(np.array([300, 301]))
    assert tok.decode(torch.tensor([300, 301])) == plain.decode(torch.tensor([300, 301]))
    assert tok.decode([300, 50365, 301, 50370], output_offsets=True) == plain.decode([300, 50365, 301, 50370], output_offsets=True)
    assert len(memo) == n
    tok.add_special_tokens({"additional_special_tokens": ["<|brandnew|>"]})
    plain.add_special_tokens({"additional_special_tokens": ["<|brandnew|>"]})
    new_id = tok.convert_tokens_to_ids("<|brandnew|>")
    for kw in ({}, {"skip_special_tokens": True}):
        assert tok.decode([300, new_id], **kw) == plain.decode([300, new_id], **kw)
    assert tok.__dict__["_tw_decode_memo"][1] is not memo     # tables changed -> fresh memo


def test_model_size_s_maps_to_fp8_only_by_deployment_switch(monkeypatch):
    """The reference's model_size="S" is its quantised engine flavour (R:thestage_speechkit/nvidia/asr_pipeline.py:47-56).  Here
    every size runs the bf16 kernels unless the deployment sets THEWHISPER_SIZE_S=fp8 (or the caller passes decoder_weights)."""
    seen = []

    def spy_factory(dims, T, max_batch, dtype, heads, dev):
        seen.append(dtype)
        return oracle_engine_factory(dims, T, max_batch, "f32", heads, dev)

    def build(size, **kw):
        from thewhisper_amd import ASRPipeline
        dims = wo.PRESETS["micro"]
        model = hr.build_hf_model(dims, wo.make_weights(dims, 0)).to(torch.bfloat16)
        return ASRPipeline(model, feature_extractor=hr.build_feature_extractor(dims, 10), tokenizer=hr.build_tokenizer(dims),
                           model_size=size, chunk_length_s=10, device="cpu", torch_dtype=torch.bfloat16, engine_factory=spy_factory, **kw)

    build("S")
    monkeypatch.setenv("THEWHISPER_SIZE_S", "fp8")
    build("S")
    build("XL")
    build("S", decoder_weights="bf16")
    assert seen == ["bf16", "fp8", "bf16", "bf16"]
