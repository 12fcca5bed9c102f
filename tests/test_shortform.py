"""thewhisper_amd/shortform.py (the restated short-form seek loop) against HF's own ``WhisperGenerationMixin.generate``.

Both run on the SAME engine (the numpy oracle behind the engine interface, tests/oracle_engine.py), so every difference
would be a difference in control flow: segment cut-out, padding removal, segment slicing at the timestamp tokens, seek
advance, time offsets of the token timestamps, padding of the batch.  Compared: the raw ``generate`` return values
(ids, token timestamps, every field of every segment) and the finished pipeline dictionaries, for all three
``return_timestamps`` modes, ragged buffers (several seek passes per chunk) and batches that shrink.
"""
import numpy as np
import pytest
import torch

from oracle import hf_reference as hr
from oracle import whisper_oracle as wo
from tests.oracle_engine import oracle_engine_factory

torch.set_grad_enabled(False)
GK = {"num_beams": 1, "do_sample": False, "use_cache": True, "language": "en", "max_new_tokens": 24}


def build(preset="micro", chunk_s=10, batch_size=4, seed=0):
    from thewhisper_amd import ASRPipeline

    dims = wo.PRESETS[preset]
    model = hr.build_hf_model(dims, wo.make_weights(dims, seed))
    return ASRPipeline(model, feature_extractor=hr.build_feature_extractor(dims, chunk_s), tokenizer=hr.build_tokenizer(dims),
                       chunk_length_s=chunk_s, device="cpu", torch_dtype=torch.float32, batch_size=batch_size,
                       engine_factory=oracle_engine_factory)


def same(a, b, path=""):
    """Deep equality of generate() results: tensors by dtype + shape + value, containers recursively."""
    if isinstance(a, torch.Tensor) or isinstance(b, torch.Tensor):
        assert isinstance(a, torch.Tensor) and isinstance(b, torch.Tensor), path
        assert a.dtype == b.dtype and a.shape == b.shape, (path, a.dtype, b.dtype, a.shape, b.shape)
        assert torch.equal(a, b), path
    elif isinstance(a, dict):
        assert isinstance(b, dict) and sorted(a.keys()) == sorted(b.keys()), (path, a.keys(), b.keys())
        for k in a:
            same(a[k], b[k], f"{path}.{k}")
    elif isinstance(a, (list, tuple)):
        assert isinstance(b, (list, tuple)) and len(a) == len(b), (path, len(a), len(b))
        for i, (x, y) in enumerate(zip(a, b)):
            same(x, y, f"{path}[{i}]")
    else:
        assert a == b, (path, a, b)


def clip_features(pipe, n_clips, chunk_s, seed0=0):
    fe = pipe.feature_extractor
    kinds = ["speechlike", "noise", "sine", "speechlike", "zeros"]
    lens = [chunk_s * 16000, 7 * 1600 * chunk_s, 16000 * chunk_s // 3, chunk_s * 16000 - 4321, chunk_s * 8000]
    pcm = [wo.synth_audio(lens[i % len(lens)], seed0 + i, kinds[i % len(kinds)]) for i in range(n_clips)]
    return fe(pcm, sampling_rate=16000, return_tensors="pt", return_attention_mask=True)


@pytest.mark.parametrize("mode,seed", [("word", 0), ("word", 1), ("segments", 0), ("none", 1)])   # (every mode, both fixtures; ~30 s each on CPU)
def test_generate_equals_hf_control_flow(mode, seed):
    pipe = build(seed=seed, chunk_s=30)     # 30 s chunks: the random-weight model's timestamps leave room for a second seek pass
    model = pipe.model
    feats = clip_features(pipe, 4, 30, seed0=10 * seed)
    kw = dict(GK)
    if mode == "word":
        kw.update(return_timestamps=True, return_token_timestamps=True, return_segments=True)
    elif mode == "segments":
        kw.update(return_timestamps=True)
    else:
        kw.update(return_timestamps=False)
    call = lambda: model.generate(input_features=feats.input_features, attention_mask=feats.attention_mask,  # noqa: E731
                                  generation_config=pipe.generation_config, **kw)
    model.fast_generate = False
    n0 = model.engine.calls["generate"]
    ref = call()                       # HF's WhisperGenerationMixin.generate, every time
    passes_ref = model.engine.calls["generate"] - n0
    model.fast_generate = True
    first = call()                     # learns the plan (still HF's flow)
    assert model.last_plan is not None, "the call should have been eligible for the short-form plan"
    n0 = model.engine.calls["generate"]
    fast = call()                      # the restated loop
    assert model.engine.calls["generate"] - n0 == passes_ref, "same number of engine passes as HF's loop"
    same(ref, first, "learn")
    same(ref, fast, "fast")
    if mode != "none":
        assert passes_ref > 1, "fixture should need several seek passes (otherwise the loop is not exercised)"


def test_pipeline_outputs_identical_with_and_without_the_fast_path():
    pipe = build(batch_size=3)
    audios = [wo.synth_audio(n, s, k) for n, s, k in [(160000, 5, "speechlike"), (250000, 6, "noise"), (52000, 7, "sine"),
                                                       (400000, 8, "speechlike")]]
    for rt in (False, True, "word"):
        outs = {}
        for fast in (False, True, True):
            pipe.model.fast_generate = fast
            outs.setdefault(fast, []).append(pipe([a.copy() for a in audios], generate_kwargs=dict(GK), chunk_length_s=9,
                                                  return_timestamps=rt, batch_size=3))
        assert outs[False][0] == outs[True][0] == outs[True][1], rt


def test_ineligible_calls_keep_hf_flow():
    pipe = build()
    model = pipe.model
    feats = clip_features(pipe, 2, 10)
    base = dict(input_features=feats.input_features, attention_mask=feats.attention_mask, generation_config=pipe.generation_config)
    for extra in (dict(prompt_ids=torch.tensor([50362, 300, 301])), dict(temperature=(0.0, 0.2), logprob_threshold=-1.0),
                  dict(condition_on_prev_tokens=True)):
        kw = {**GK, "return_timestamps": True, **extra}
        assert model._plan_key({**base, **kw}) is None, extra
    assert model._plan_key({**base, **GK, "return_timestamps": True}) is not None
    # language detection (no language given on a multilingual model) is per-row: not eligible
    kw = {k: v for k, v in GK.items() if k != "language"}
    assert model._plan_key({**base, **kw, "return_timestamps": True}) is None


def test_run_pass_accepts_chunks_from_different_calls():
    """The property the serving hub relies on: chunks at different seek positions (and from different buffers) share one pass,
    and each ends with exactly what it gets when decoded alone."""
    from thewhisper_amd import shortform

    pipe = build(batch_size=4, chunk_s=30)
    model = pipe.model
    feats = clip_features(pipe, 4, 30, seed0=40)
    kw = dict(GK, return_timestamps=True, return_token_timestamps=True, return_segments=True)
    call = lambda f, m: model.generate(input_features=f, attention_mask=m, generation_config=pipe.generation_config, **kw)  # noqa: E731
    call(feats.input_features, feats.attention_mask)                 # learn
    plan = model.last_plan
    alone = [call(feats.input_features[i : i + 1], feats.attention_mask[i : i + 1]) for i in range(4)]
    nf = [int(x) for x in feats.attention_mask.sum(-1)]
    works = [shortform.ChunkWork(feats.input_features[i], nf[i], tag=i) for i in range(4)]
    eng = model.engine
    shortform.run_pass(eng, plan, works[:2])                         # two chunks get a head start ...
    pending = [w for w in works if not w.done]
    while pending:                                                   # ... then everything unfinished shares the passes
        shortform.run_pass(eng, plan, pending)
        pending = [w for w in works if not w.done]
    assert max(w.passes for w in works) > 1
    for i, w in enumerate(works):
        seq, raw, seg = shortform.work_tokens(plan, w)
        same(seq, alone[i]["sequences"][0], f"ids[{i}]")
        same(raw, alone[i]["token_timestamps"][0], f"ts[{i}]")
        same(seg, torch.cat([s["token_timestamps"] for s in alone[i]["segments"][0]]), f"segment ts[{i}]")


def test_pass_assembled_in_groups_equals_one_group():
    """shortform.Pass: chunks added as two groups (the hub's "leftovers first, late arrivals behind them") end exactly like the same
    chunks decoded as one group - the engine's slot-offset entry points (tw_encode_at / tw_cross_kv_at) are used for the second group."""
    from thewhisper_amd import shortform

    pipe = build(batch_size=4)
    model = pipe.model
    feats = clip_features(pipe, 4, 10, seed0=70)
    kw = dict(GK, return_timestamps=True, return_token_timestamps=True, return_segments=True)
    model.generate(input_features=feats.input_features, attention_mask=feats.attention_mask, generation_config=pipe.generation_config, **kw)
    plan, eng = model.last_plan, model.engine
    nf = [int(x) for x in feats.attention_mask.sum(-1)]

    def decode(groups):
        works = [shortform.ChunkWork(feats.input_features[i], nf[i]) for i in range(4)]
        p = shortform.Pass(eng, plan)
        for g in groups:
            p.add([works[i] for i in g])
        assert p.free == 0
        p.run()
        return [shortform.work_tokens(plan, w) for w in works]

    n0 = eng.calls["encode"]
    one = decode([[0, 1, 2, 3]])
    two = decode([[0, 1, 2], [3]])
    three = decode([[0], [1, 2], [3]])
    assert eng.calls["encode"] - n0 == 1 + 2 + 3
    for a, b, c in zip(one, two, three):
        same(list(a), list(b), "two groups")
        same(list(a), list(c), "three groups")
    with pytest.raises(ValueError):
        p = shortform.Pass(eng, plan)
        p.add([shortform.ChunkWork(feats.input_features[i % 4], nf[i % 4]) for i in range(5)])


def test_call_wider_than_the_engine_and_a_decoder_that_never_advances():
    """generate_shortform on more rows than the engine holds runs several passes per seek iteration with unchanged results;
    a decoder that closes every segment at <|0.00|> (seek never advances - HF's loop would spin forever) raises."""
    from thewhisper_amd import shortform

    pipe = build(batch_size=4)
    model = pipe.model
    feats = clip_features(pipe, 4, 10, seed0=90)
    kw = dict(GK, return_timestamps=True, return_token_timestamps=True, return_segments=True)
    ref = model.generate(input_features=feats.input_features, attention_mask=feats.attention_mask, generation_config=pipe.generation_config, **kw)
    plan, eng = model.last_plan, model.engine

    class Narrow:      # the same engine with room for 3 rows
        max_batch = 3

        def __getattr__(self, k):
            return getattr(eng, k)

    got = shortform.generate_shortform(Narrow(), plan, feats.input_features, feats.attention_mask)
    same(got["sequences"], ref["sequences"], "ids")
    same(got["token_timestamps"], ref["token_timestamps"], "ts")

    tb = plan.timestamp_begin

    class Stuck(Narrow):
        def generate_greedy(self, prompt, **kw):
            B = prompt.shape[0]
            tail = np.tile(np.array([[tb, 7, tb, tb, 9]], dtype=np.int32), (B, 1))   # "<|0.00|> w <|0.00|><|0.00|> w": seek += 0
            seq = np.concatenate([np.asarray(prompt, np.int32), tail], axis=1)
            return {"sequences": seq, "length": seq.shape[1]}

        def token_timestamps(self, B, n_prompt, L, nf, tp):
            return np.zeros((B, L), np.float32)

    with pytest.raises(RuntimeError, match="seek passes"):
        shortform.generate_shortform(Stuck(), plan, feats.input_features[:2], feats.attention_mask[:2])


def test_kept_columns_rule_of_the_host_mirror_equals_the_pinned_restatement():
    """`shortform.hf_kept_columns` (product) and `oracle.whisper_oracle.hf_kept_columns` (pinned to HF's own `_extract_token_timestamps`
    for every flavour of `num_frames`, tests/test_oracle_vs_hf.py) are two statements of the same rule; the engine bound derived from it
    (`columns_as_num_frames`: one Python-slice crop per row, include/thewhisper.h) keeps exactly those columns."""
    import torch

    from oracle import whisper_oracle as wo
    from thewhisper_amd import shortform as sf

    rng = np.random.default_rng(0)
    cases = [None, 3000, 1067, -1721, 0, -1, 1, [3000] * 3, [-1721], [-400, -400], [-2999, -2999], [-3000] * 2, [-5000], [1067, 2900], [-1721, 900, 40],
             np.array([-1000, -1000]), torch.tensor([-400, 700]), torch.tensor([-1721])]
    cases += [rng.integers(-3200, 3200, size=int(rng.integers(1, 5))).tolist() for _ in range(200)]
    cases += [[int(rng.integers(-3200, 3200))] * int(rng.integers(1, 4)) for _ in range(200)]
    for T in (1500, 500, 100, 7):
        for nf in cases:
            B = 3 if nf is None or isinstance(nf, int) else len(nf)
            want = wo.hf_kept_columns(nf, B, T)
            got = sf.hf_kept_columns(nf, B, T)
            assert got == want, (T, nf, got, want)
            # the C ABI's per-row rule applied to the derived bound keeps exactly those columns
            for c, n in zip(got, sf.columns_as_num_frames(got)):
                k = n // 2
                assert (min(T, k) if k >= 0 else max(0, T + k)) == c
