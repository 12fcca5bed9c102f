"""Round 6 (`-m gpu`, through the C ABI):

* a stream's results do not depend on the other rows of its launch - 1, 16, 17 and 64 rows, an ordinary step or the rows mode of
  the batched prefill - BIT FOR BIT, in every context dtype, at the REAL width (d = 1280, ffn = 5120: the long-K projection takes
  16 wavefronts per tile with one group of streams and 8 with several; k_decode.hip gives both the same sixteen K slices and the
  same order of additions);
* tw_greedy_opts::n_draft (draft-and-verify): whatever is offered as a draft - the true continuation, a corrupted
  one, rubbish - the call returns what it returns without it: ids against the oracle's `greedy_generate` (strict f32) and ids,
  alignment rows and token timestamps against the engine's own plain call (every dtype), with the expected number of confirmed tokens;
* `AMDWhisperBackend(draft_previous_tick=True)` on the reference scheduler's call sequence: every call's words identical to the
  plain backend's."""
import numpy as np
import pytest
import torch

from oracle import whisper_oracle as wo
from tests.util import PROMPT, clips, dims_variant, make_engine

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("needs the MI355X")


REAL = dict(enc_layers=1, dec_layers=2)


@pytest.mark.parametrize("dtype", ["bf16", "f16", "f32", "fp8a16"])
def test_rows_of_a_launch_do_not_matter(dtype):
    """The same clip (slot 0) teacher-forced through 7 positions alone, among 16, 17 and 64 rows, and again from another slot of
    the 64-row launch: identical logits, bit for bit.  Then the rows mode: a forced-prefix call (positions in launches of up to 64
    rows) must leave exactly the alignment rows and produce exactly the ids of the step-by-step call."""
    dims = dims_variant("large-v3", **REAL)
    w = wo.make_weights(dims, 2)
    heads = [(1, 0), (1, 3)]
    T, Bmax = 100, 64
    eng = make_engine(dims, w, T=T, max_batch=Bmax, dtype=dtype, heads=heads, use_graph=False)
    try:
        pcm = clips(T * 320, Bmax)
        pcm[40] = pcm[0]                                      # the same audio in another group of 16 rows
        mel = eng.logmel(torch.from_numpy(pcm).cuda())
        ids = np.concatenate([np.tile(np.array(PROMPT), (Bmax, 1)), np.random.default_rng(3).integers(0, 50000, size=(Bmax, 4))], axis=1)
        ids[40] = ids[0]

        def run(B):
            eng.encode(mel[:B]); eng.cross_kv(B); eng.decoder_reset(B)
            return np.stack([eng.decode_step(ids[:B, s].tolist()).cpu().numpy() for s in range(ids.shape[1])], axis=1)

        one = run(1)
        for B in (16, 17, 64, 33):
            got = run(B)
            assert np.array_equal(got[0], one[0]), f"{dtype}: row 0 of a {B}-row launch differs from the one-row launch"
            if B == 64:
                assert np.array_equal(got[40], one[0]), f"{dtype}: slot 40 differs from slot 0 on the same audio"
        # rows mode (prefill launches of 16-64 rows) against the one-position steps: free-running ids, alignment rows, timestamps
        for B in (1, 3):
            eng.encode(mel[:B]); eng.cross_kv(B)
            prompt = np.tile(np.array(PROMPT, dtype=np.int32), (B, 1))
            kw = dict(max_new_tokens=40, min_new_tokens=40, timestamps=True, want_alignment=True)
            full = eng.generate_greedy(prompt, **kw)
            L = full["length"]
            al_full = eng.get_alignment(B, L - 1)
            forced = full["sequences"][:, : 3 + 30].astype(np.int32)
            out = eng.generate_greedy(forced, n_forced=30, **kw)
            assert np.array_equal(out["sequences"], full["sequences"]), dtype
            assert np.array_equal(eng.get_alignment(B, L - 1), al_full), dtype
    finally:
        eng.close()


@pytest.mark.parametrize("dtype,T", [("bf16", 500), ("f16", 1000), ("bf16", 1500), ("f32", 500)])
def test_streams_of_a_pass_do_not_matter_encoder_included(dtype, T):
    """The encoder side of the same statement: a clip's encoder states - and with them its logits - are bit for bit the same whether it
    is encoded alone (M = T rows: 64 x 64 tiles, plain epilogue), with two others, or among 16 / 40 clips (M >= 8000 rows: the
    large-M kernel with its LDS-staged epilogue, 128-query attention workgroups).  Until round 6 the two epilogues summed the
    LayerNorm statistics they leave for the next GEMM in different orders (18 % of the 16-bit encoder states differed between a
    one-clip and a 16-clip pass); tw_common.h: tw_stat4."""
    dims = dims_variant("large-v3", enc_layers=2, dec_layers=1)
    w = wo.make_weights(dims, 2)
    eng = make_engine(dims, w, T=T, max_batch=40, dtype=dtype, heads=[(0, 0)], use_graph=False)
    try:
        mel = eng.logmel(torch.from_numpy(clips(T * 320, 40)).cuda())
        ids = np.array(PROMPT + [1234, 777])
        ref = None
        for B in (1, 3, 16, 40):
            enc = eng.encode(mel[:B], return_hidden=True)[0].float().cpu().numpy()
            eng.cross_kv(B); eng.decoder_reset(B)
            lg = np.stack([eng.decode_step([int(t)] * B).cpu().numpy()[0] for t in ids])
            if ref is None:
                ref = (enc, lg)
                continue
            assert np.array_equal(enc, ref[0]), f"{dtype} T={T}: encoder states of clip 0 among {B} clips differ from the one-clip pass"
            assert np.array_equal(lg, ref[1]), f"{dtype} T={T}: logits of clip 0 among {B} clips differ from the one-clip pass"
    finally:
        eng.close()


def _corrupt(seq, n0, where, rng, V=50000):
    d = seq[:, n0:].copy()
    for b in range(d.shape[0]):
        for j in where:
            if j < d.shape[1]:
                d[b, j] = (int(d[b, j]) + 1 + int(rng.integers(0, V - 2))) % V
    return d


DRAFT_CASES = [
    # preset, layers, T, B, max_new, forced length?, dtype, graph
    ("micro", None, 100, 1, 60, True, "f32", True), ("micro", None, 100, 1, 60, False, "f32", False), ("micro", None, 500, 3, 40, True, "f32", True),
    ("micro", None, 100, 16, 30, True, "f32", True), ("micro", None, 100, 1, 120, True, "bf16", True), ("micro", None, 750, 2, 40, True, "fp8a16", True),
    ("micro", None, 100, 5, 90, True, "f16", False), ("micro", None, 100, 2, 40, True, "fp8a8", True),
    ("large-v3", REAL, 100, 1, 100, True, "bf16", True), ("large-v3", REAL, 100, 1, 100, False, "f16", True), ("large-v3", REAL, 100, 2, 48, True, "f32", True),
    ("large-v3", REAL, 100, 4, 48, True, "fp8a16", True), ("large-v3", REAL, 100, 20, 24, True, "bf16", True),
]


@pytest.mark.parametrize("preset,layers,T,B,max_new,fixed_len,dtype,graph", DRAFT_CASES)
def test_draft_and_verify_returns_the_plain_result(preset, layers, T, B, max_new, fixed_len, dtype, graph):
    dims = dims_variant(preset, **layers) if layers else wo.PRESETS[preset]
    w = wo.make_weights(dims, 0 if layers is None else 2)
    heads = [(dims.dec_layers - 1, 0), (dims.dec_layers - 1, 1)]
    eng = make_engine(dims, w, T=T, max_batch=B, dtype=dtype, heads=heads, use_graph=graph)
    try:
        pcm = clips(T * 320, B)
        mel = eng.logmel(torch.from_numpy(pcm).cuda(), out_dtype=torch.float32)
        eng.encode(mel); eng.cross_kv(B)
        prompt = np.tile(np.array(PROMPT, dtype=np.int32), (B, 1))
        kw = dict(max_new_tokens=max_new, min_new_tokens=max_new if fixed_len else 0, timestamps=True, want_alignment=True)
        full = eng.generate_greedy(prompt, **kw)
        L = full["length"]
        seq = full["sequences"]
        al_full = eng.get_alignment(B, L - 1)
        nf = [2 * T] * B
        ts_full = eng.token_timestamps(B, 3, L, nf)
        if dtype == "f32" and preset == "micro":
            om = wo.OracleWhisper(dims, w, T=T)
            opt = wo.GreedyOptions(max_new_tokens=max_new, min_new_tokens=max_new if fixed_len else 0, timestamps=True, alignment_heads=heads)
            ref = wo.greedy_generate(om, om.encode(wo.log_mel(pcm, dims.n_mels)), prompt, opt)
            assert np.array_equal(seq, ref["sequences"])
        # what may be offered: generated tokens before any <eos> / padding of any row
        gen = seq[:, 3:]
        n_ok = int(min([np.nonzero(r == 50257)[0][0] if (r == 50257).any() else len(r) for r in gen]))
        n_ok = min(n_ok, max_new - 2)
        assert n_ok >= 4, "the case needs a few tokens to offer"
        rng = np.random.default_rng(5)
        variants = {
            "true continuation": (gen[:, :n_ok].copy(), n_ok),
            "half of it": (gen[:, : n_ok // 2].copy(), n_ok // 2),
            "one token": (gen[:, :1].copy(), 1),
            "wrong from the start": (_corrupt(seq[:, : 3 + n_ok], 3, range(n_ok), rng), None),
            "one wrong token": (_corrupt(seq[:, : 3 + n_ok], 3, [n_ok // 3], rng), None),
            "three wrong tokens": (_corrupt(seq[:, : 3 + n_ok], 3, [2, n_ok // 2, n_ok - 1], rng), None),
        }
        if B > 1:       # only ONE stream's draft is wrong: the round ends there for everybody, the others' drafts are offered again
            d = gen[:, :n_ok].copy()
            d[B - 1, n_ok // 2] = (int(d[B - 1, n_ok // 2]) + 7) % 50000
            variants["one stream wrong"] = (d, None)
        for name, (draft, want_acc) in variants.items():
            draft = np.where(draft == 50257, 0, draft).astype(np.int32)
            out = eng.generate_greedy(np.concatenate([prompt, draft], axis=1), n_draft=draft.shape[1], **kw)
            what = f"{preset} {dtype} B={B}: draft = {name}"
            assert out["length"] == L and np.array_equal(out["sequences"], seq), what
            assert np.array_equal(eng.get_alignment(B, L - 1), al_full), what
            assert np.array_equal(eng.token_timestamps(B, 3, L, nf), ts_full), what
            dr = out["draft"]
            assert dr["offered"] == draft.shape[1] * B and dr["launches"] >= 1 and dr["rounds"] >= 1
            n_same = int(np.min([np.argmax(np.append(draft[b] != gen[b, : draft.shape[1]], True)) for b in range(B)]))
            assert dr["accepted"] >= n_same * B, (what, dr, n_same)          # round 1 confirms the common prefix, later rounds may add
            if want_acc is not None:
                assert dr["accepted"] == want_acc * B and dr["rounds"] == 1, (what, dr)
            if name == "one wrong token" and B <= 4:       # the rest of the draft is confirmed by the rounds that follow
                assert dr["accepted"] == (n_ok - 1) * B and dr["rounds"] >= 2, (what, dr)
        # misuse
        with pytest.raises(RuntimeError, match="n_draft"):
            eng.generate_greedy(np.concatenate([prompt, gen[:, :4].astype(np.int32)], axis=1), n_draft=4, n_forced=2, **kw)
        with pytest.raises(RuntimeError, match="n_draft"):
            eng.generate_greedy(prompt, n_draft=3, **kw)
        bad = np.concatenate([prompt, gen[:, :4].astype(np.int32)], axis=1); bad[0, -1] = 50257
        with pytest.raises(RuntimeError, match="draft token"):
            eng.generate_greedy(bad, n_draft=4, **kw)
        # and the context is as usable as before
        again = eng.generate_greedy(prompt, **kw)
        assert np.array_equal(again["sequences"], seq)
    finally:
        eng.close()


def test_backend_with_drafts_reproduces_the_plain_backend_on_the_scheduler_sequence():
    """`AMDWhisperBackend(draft_previous_tick=True)` on the golden stream's call sequence (ragged rolling buffers as the reference
    scheduler sends them): every call returns the plain backend's words - text, start, end - exactly."""
    import json
    import os

    from tests.test_pipeline_glue import build_amd_pipeline, normalise
    from thewhisper_amd import AMDWhisperBackend

    gold = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "pipeline_golden.json")))["streaming_micro_c10"]
    audio = wo.synth_audio(16000 * gold["seconds"], gold["seed"], gold["kind"])
    plain = AMDWhisperBackend(None, chunk_length_s=10, asr_pipeline=build_amd_pipeline("micro", 10, 1, device="cuda", engine_factory=None),
                              draft_previous_tick=False)
    draft = AMDWhisperBackend(None, chunk_length_s=10, asr_pipeline=build_amd_pipeline("micro", 10, 1, device="cuda", engine_factory=None),
                              draft_previous_tick=True)
    for c in gold["calls"]:
        buf = audio[c["offset"] : c["offset"] + c["n"]]
        a = plain.transcribe(buf.copy(), c["t0"], 16000)
        b = draft.transcribe(buf.copy(), c["t0"], 16000)
        assert normalise(a) == normalise(b), c
        assert [(w["start"], w["end"]) for w in a] == [(w["start"], w["end"]) for w in b], c
    st = draft.reuse_stats
    print(f"\nDRAFT (MI355X, micro model, golden stream): {st['reused']} of {st['calls']} calls offered a draft, {st['draft_tokens']} tokens offered, "
          f"{st['confirmed_tokens']} confirmed, {st['verify_launches']} verify launches")
    assert st["reused"] >= 1 and st["draft_tokens"] > 0


@pytest.mark.parametrize("dtype,layers,n_calls", [("f16", 2, 60), ("bf16", 2, 60), ("f16", 32, 30)])
def test_config3_trace_at_the_real_width_draft_ids_equal_plain_ids(dtype, layers, n_calls):
    """BASELINE config 3's call pattern (tests/golden/config3_trace.json: the rolling buffers the REFERENCE'S scheduler hands its
    backend for a 60 s stream, R:thestage_speechkit/streaming/streaming_pipeline.py:388-435, :770-796) at whisper-large-v3's width
    (d = 1280, ffn = 5120, 20 heads, full vocabulary; 2 + 2 layers so that the case takes seconds, and the first 30 calls at the
    FULL 32 + 32 layers in float16): the first calls of the trace through the
    plain backend and through `draft_previous_tick=True` on the same engine - the token ids the engine decoded (every seek pass,
    before tokenizer / filter / merge) are IDENTICAL call by call, the words too, and drafts were offered and confirmed.  What
    bench.py reports for the full depth (`config3.with_draft_previous_tick.token_identity_mean` = 1.0) is asserted here."""
    import json
    import os

    import bench
    from thewhisper_amd import AMDWhisperBackend, ASRPipeline, synthetic
    from thewhisper_amd.engine import WhisperEngine
    from transformers import WhisperFeatureExtractor

    dims = dict(bench.DIMS["large-v3"], enc_layers=layers, dec_layers=layers)
    heads = [(layers - 1, 0), (layers - 1, 3)]
    trace = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "config3_trace.json")))
    chunk_s = trace["chunk_length_s"]
    tdt = {"f16": torch.float16, "bf16": torch.bfloat16}[dtype]
    eng = WhisperEngine(dims, 50 * chunk_s, max_batch=1, dtype=dtype, alignment_heads=heads, use_graph=True)
    try:
        eng.load_state_dict(bench.random_state_dict(dims, torch.device("cuda", 0), seed=0))
        model = synthetic.skeleton_model(dims, device="cuda:0", dtype=tdt, alignment_heads=heads)
        pipe = ASRPipeline(model, feature_extractor=WhisperFeatureExtractor(feature_size=dims["n_mels"], chunk_length=chunk_s),
                           tokenizer=synthetic.build_tokenizer(dims["vocab"]), chunk_length_s=chunk_s, device="cuda:0", torch_dtype=tdt,
                           batch_size=1, engine=eng)
        stream = (np.random.default_rng(trace["seed"]).standard_normal(16000 * trace["seconds"]) * 0.1).clip(-1, 1).astype(np.float32)
        seen = []
        inner = eng.generate_greedy

        def recording(prompt, **kw):
            out = inner(prompt, **kw)
            for row in out["sequences"][:, 3:]:
                hit = np.nonzero(row == int(kw.get("eos_id", 50257)))[0]
                seen.append(np.asarray(row[: int(hit[0])] if len(hit) else row, dtype=np.int64))
            return out

        eng.generate_greedy = recording
        got = {}
        for variant in ("plain", "draft"):
            be = AMDWhisperBackend(None, chunk_length_s=chunk_s, asr_pipeline=pipe, draft_previous_tick=(variant == "draft"))
            be.transcribe(stream[:16000], 0.0, 16000)      # plan learning outside the compared calls
            be.reset()
            ids, words = [], []
            for c in trace["calls"][:n_calls]:
                seen.clear()
                words.append(be.transcribe(stream[c["offset"] : c["offset"] + c["n"]], c["t0"], 16000))
                ids.append(np.concatenate(seen) if seen else np.zeros(0, np.int64))
            got[variant] = (ids, words, dict(be.reuse_stats))
        for i, (a, b) in enumerate(zip(got["plain"][0], got["draft"][0])):
            assert len(a) == len(b) and (a == b).all(), f"{dtype}: call {i} of the trace decodes other ids with a draft"
        assert got["plain"][1] == got["draft"][1]
        st = got["draft"][2]
        print(f"\nCONFIG3 {dtype} (large-v3 width, {layers} + {layers} layers, {n_calls} calls): {st['reused']} calls offered a draft, {st['draft_tokens']} tokens offered, "
              f"{st['confirmed_tokens']} confirmed, {st['verify_launches']} verify launches; ids identical on every call")
        assert st["reused"] >= n_calls // 2 and st["confirmed_tokens"] > 0
    finally:
        eng.generate_greedy = inner
        eng.close()


@pytest.mark.parametrize("dtype", ["f32", "bf16", "f16", "fp8a16", "fp8a8"])
def test_draft_fuzz_random_drafts_never_change_the_result(dtype):
    """Property test of `n_draft` on the micro model: 60 random situations per dtype - 1 ... 16 streams, 6 ... 90 new tokens, fixed or
    free length (<eos> allowed early), timestamp grammar on or off, a suppress list or none, drafts of random length whose tokens are
    corrupted at a random rate (0 = the true continuation ... 1 = rubbish, sometimes in one stream only) - and in every one of them the
    call returns the plain call's ids, length, alignment rows and token timestamps; graph replay on and off."""
    dims = wo.PRESETS["micro"]
    w = wo.make_weights(dims, 0)
    heads = [(1, 0), (1, 1)]
    T = 100
    rng = np.random.default_rng({"f32": 1, "bf16": 2, "f16": 3, "fp8a16": 4, "fp8a8": 5}[dtype])
    engines = {g: make_engine(dims, w, T=T, max_batch=16, dtype=dtype, heads=heads, use_graph=g) for g in (True, False)}
    try:
        pcm = clips(T * 320, 16)
        n_engaged = 0
        for case in range(60):
            eng = engines[bool(case % 2)]
            B = int(rng.choice([1, 1, 2, 3, 5, 16]))
            sel = rng.permutation(16)[:B]
            mel = eng.logmel(torch.from_numpy(pcm[sel]).cuda(), out_dtype=torch.float32)
            eng.encode(mel); eng.cross_kv(B)
            max_new = int(rng.integers(6, 91))
            ts_on = bool(rng.integers(0, 2))
            kw = dict(max_new_tokens=max_new, min_new_tokens=max_new if rng.integers(0, 2) else 0, timestamps=ts_on, want_alignment=True)
            if rng.integers(0, 3) == 0:
                kw["suppress"] = tuple(int(x) for x in rng.integers(0, 50000, size=20))
            prompt = np.tile(np.array(PROMPT if ts_on else PROMPT + [50364], dtype=np.int32), (B, 1))
            n0 = prompt.shape[1]
            full = eng.generate_greedy(prompt, **kw)
            L, seq = full["length"], full["sequences"]
            al = eng.get_alignment(B, L - 1)
            ts = eng.token_timestamps(B, n0, L, [2 * T] * B)
            gen = seq[:, n0:]
            n_ok = int(min([np.nonzero(r == 50257)[0][0] if (r == 50257).any() else len(r) for r in gen]))
            n_ok = min(n_ok, max_new - 2)
            if n_ok < 1:
                continue
            nd = int(rng.integers(1, n_ok + 1))
            draft = gen[:, :nd].copy().astype(np.int32)
            rate = float(rng.choice([0.0, 0.0, 0.03, 0.15, 0.5, 1.0]))
            rows = range(B) if rng.integers(0, 2) else [int(rng.integers(0, B))]
            for b in rows:
                for j in range(nd):
                    if rng.random() < rate:
                        draft[b, j] = int(rng.integers(0, 50000))
            draft = np.where(draft == 50257, 0, draft).astype(np.int32)
            out = eng.generate_greedy(np.concatenate([prompt, draft], axis=1), n_draft=nd, **kw)
            what = f"{dtype} case {case}: B={B} max_new={max_new} ts={ts_on} nd={nd} rate={rate} graph={bool(case % 2)}"
            assert out["length"] == L and np.array_equal(out["sequences"], seq), what
            assert np.array_equal(eng.get_alignment(B, L - 1), al), what
            assert np.array_equal(eng.token_timestamps(B, n0, L, [2 * T] * B), ts), what
            n_engaged += 1
        assert n_engaged >= 40
    finally:
        for e in engines.values():
            e.close()


@pytest.mark.parametrize("dtype", ["bf16", "f16"])
def test_batch_composition_fuzz_at_the_real_width(dtype):
    """Property form of "who else is in the batch does not matter" (large-v3 width, 1 + 2 layers, T = 500 so that a 16-clip pass reaches
    the large-M encoder kernel): 12 random passes - 1 ... 40 clips drawn at random, in random slots - and in every one of them each
    clip's encoder states and teacher-forced logits are bit for bit those of the clip alone."""
    dims = dims_variant("large-v3", enc_layers=1, dec_layers=2)
    w = wo.make_weights(dims, 2)
    T, N = 500, 40
    eng = make_engine(dims, w, T=T, max_batch=N, dtype=dtype, heads=[(1, 0)], use_graph=False)
    try:
        mel = eng.logmel(torch.from_numpy(clips(T * 320, N)).cuda())
        ids = PROMPT + [4242, 99]
        rng = np.random.default_rng(7)

        def run(sel):
            B = len(sel)
            enc = eng.encode(mel[torch.as_tensor(sel)], return_hidden=True).float().cpu().numpy()
            eng.cross_kv(B); eng.decoder_reset(B)
            lg = np.stack([eng.decode_step([int(t)] * B).cpu().numpy() for t in ids], axis=1)
            return enc, lg

        alone = {}
        for case in range(12):
            B = int(rng.choice([1, 2, 3, 7, 15, 16, 17, 31, 33, 40]))
            sel = rng.permutation(N)[:B].tolist()
            enc, lg = run(sel)
            for slot in {0, B - 1, int(rng.integers(0, B))}:
                c = sel[slot]
                if c not in alone:
                    alone[c] = run([c])
                    enc, lg = run(sel)      # (the one-clip pass overwrote the context)
                assert np.array_equal(enc[slot], alone[c][0][0]), f"{dtype}: clip {c} in slot {slot} of a {B}-clip pass: encoder states differ"
                assert np.array_equal(lg[slot], alone[c][1][0]), f"{dtype}: clip {c} in slot {slot} of a {B}-clip pass: logits differ"
    finally:
        eng.close()
