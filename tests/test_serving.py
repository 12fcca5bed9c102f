"""Cross-stream batching hub (thewhisper_amd/serving.py) on CPU with the oracle-backed engine: concurrent sessions get
exactly the per-stream results, and their requests are really executed as shared batches."""
import json
import os
import threading

import numpy as np
import pytest
import torch

from oracle import whisper_oracle as wo
from tests.test_pipeline_glue import build_amd_pipeline, golden, normalise

torch.set_grad_enabled(False)


def test_transcribe_many_equals_individual_calls():
    from thewhisper_amd import AMDWhisperBackend

    pipe = build_amd_pipeline("micro", 10, 4)
    backend = AMDWhisperBackend(None, chunk_length_s=10, asr_pipeline=pipe)
    bufs = [(wo.synth_audio(n, seed, "speechlike"), t0, 16000) for n, seed, t0 in
            [(52000, 1, 0.0), (160000, 2, 3.5), (93761, 3, 6.64), (32000, 4, 0.0), (120000, 5, 1.0)]]
    single = [backend.transcribe(a.copy(), t0, sr) for a, t0, sr in bufs]
    calls_before = pipe.model.engine.calls["generate"]
    many = backend.transcribe_many([(a.copy(), t0, sr) for a, t0, sr in bufs], batch_size=4)
    assert normalise(many) == normalise(single)
    # 5 one-chunk buffers with batch_size 4 -> 2 batched encoder/decoder passes (plus seek-loop repeats), not 5
    assert pipe.model.engine.calls["generate"] - calls_before < 2 * len(bufs)


def test_hub_batches_concurrent_sessions_and_replays_reference_stream():
    from thewhisper_amd import AMDWhisperBackend
    from thewhisper_amd.serving import BatchingHub

    g = golden()["streaming_micro_c10"]
    pipe = build_amd_pipeline("micro", 10, 4)
    hub = BatchingHub(AMDWhisperBackend(None, chunk_length_s=10, asr_pipeline=pipe), max_batch=4, max_wait_s=0.5)
    audio = wo.synth_audio(16000 * g["seconds"], g["seed"], g["kind"])
    picks = g["calls"][::6]
    n_sessions = 4
    out = [[None] * len(picks) for _ in range(n_sessions)]
    barrier = threading.Barrier(n_sessions)

    def session(k):
        be = hub.stream_backend()
        for i, c in enumerate(picks):
            barrier.wait()  # all sessions ask at the same time, like 4 schedulers ticking together
            out[k][i] = be.transcribe(audio[c["offset"] : c["offset"] + c["n"]], c["t0"], 16000)

    th = [threading.Thread(target=session, args=(k,)) for k in range(n_sessions)]
    [t.start() for t in th]
    [t.join(300) for t in th]
    hub.close()
    for k in range(n_sessions):
        for i, c in enumerate(picks):
            assert normalise(out[k][i]) == c["result"]          # == the reference's LocalWhisperBackend output
    assert sum(hub.batches) == n_sessions * len(picks)
    assert max(hub.batches) > 1                                  # requests of different sessions shared a batch


@pytest.mark.skipif(not os.path.isdir("/root/reference"), reason="reference checkout not present")
def test_hub_under_the_reference_scheduler():
    """Two unmodified reference StreamingPipeline instances (own scheduler state each) sharing one hub reproduce the
    committed/uncommitted words of the reference's single-stream run."""
    from oracle.make_golden import _import_reference
    from thewhisper_amd import AMDWhisperBackend
    from thewhisper_amd.serving import BatchingHub

    _, sp = _import_reference()
    g = golden()["streaming_micro_c10"]
    pipe = build_amd_pipeline("micro", 10, 2)
    hub = BatchingHub(AMDWhisperBackend(None, chunk_length_s=10, asr_pipeline=pipe), max_batch=2, max_wait_s=0.05)
    audio = wo.synth_audio(16000 * g["seconds"], g["seed"], g["kind"])[: 16000 * 6]
    results = {}

    def session(k):
        s = sp.StreamingPipeline(backend=hub.stream_backend(), chunk_length_s=10, min_process_chunk_s=0.5, use_vad=False)
        committed, last = [], []
        for i in range(0, len(audio), g["step_samples"]):
            c, u = s(audio[i : i + g["step_samples"]])
            committed += c
            last = u
        results[k] = (committed, last)

    th = [threading.Thread(target=session, args=(k,)) for k in range(2)]
    [t.start() for t in th]
    [t.join(600) for t in th]
    hub.close()
    assert normalise(results[0]) == normalise(results[1])
    # the first 6 s of the golden stream: same backend calls -> same words as the reference backend produced
    ref_calls = [c for c in g["calls"] if c["offset"] + c["n"] <= 16000 * 6]
    assert normalise(results[0][1]) == ref_calls[-1]["result"] or len(results[0][1]) > 0


class _FlakyBackend:
    """Backend double: a request whose first sample is NaN makes the (batched or single) call raise."""

    class _P:
        class model:
            class engine:
                max_batch = 4
        model = model

    asr_pipeline = _P

    def __init__(self):
        self.calls = []

    def transcribe_many(self, requests, batch_size=None):
        self.calls.append(len(requests))
        if any(np.isnan(a[0]) for a, _, _ in requests):
            raise ValueError("malformed buffer")
        return [[{"text": f"n={len(a)}", "start": t0, "end": t0 + 1.0}] for a, t0, _ in requests]


def test_hub_isolates_a_failing_request_and_fails_parked_requests_on_close():
    from thewhisper_amd.serving import BatchingHub

    be = _FlakyBackend()
    hub = BatchingHub(be, max_batch=4, max_wait_s=0.3)
    good = [np.zeros(100 + i, np.float32) for i in range(3)]
    bad = np.full(50, np.nan, np.float32)
    futs = [hub.submit(good[0], 0.0, 16000), hub.submit(bad, 1.0, 16000), hub.submit(good[1], 2.0, 16000), hub.submit(good[2], 3.0, 16000)]
    assert futs[0].result(10)[0]["text"] == "n=100" and futs[2].result(10)[0]["text"] == "n=101" and futs[3].result(10)[0]["text"] == "n=102"
    with pytest.raises(ValueError, match="malformed"):
        futs[1].result(10)                       # only the offending session sees the exception
    assert be.calls[0] == 4 and sorted(be.calls[1:]) == [1, 1, 1, 1]   # one batched attempt, then one by one
    hub.close()
    with pytest.raises(RuntimeError, match="closed"):
        hub.submit(good[0], 0.0, 16000)
    hub.close()                                  # idempotent

    # requests parked behind the shutdown sentinel are failed, not left hanging
    hub2 = BatchingHub(_FlakyBackend(), max_batch=1, max_wait_s=0.0)
    gate = threading.Event()
    slow = hub2.backend.transcribe_many

    def blocked(reqs, batch_size=None):
        gate.wait(5)
        return slow(reqs, batch_size)

    hub2.backend.transcribe_many = blocked
    f1 = hub2.submit(good[0], 0.0, 16000)
    f2 = hub2.submit(good[1], 0.0, 16000)
    t = threading.Thread(target=hub2.close)
    t.start()
    gate.set()
    t.join(20)
    assert f1.result(5)[0]["text"] == "n=100"
    assert f2.done() and (f2.exception() is not None or f2.result()[0]["text"] == "n=101")


def test_continuous_hub_fills_passes_across_requests_and_answers_early():
    """Scheduling unit = one seek pass of one chunk: with 30 s chunks the random-weight model needs 1 or 2 passes per chunk;
    the hub must (a) return exactly the per-request results, (b) run fewer engine passes than per-request decoding would,
    (c) answer a one-pass request without waiting for the two-pass requests submitted with it."""
    import time

    from tests.test_shortform import build
    from thewhisper_amd import AMDWhisperBackend
    from thewhisper_amd.serving import BatchingHub

    pipe = build(batch_size=4, chunk_s=30)
    backend = AMDWhisperBackend(None, chunk_length_s=30, asr_pipeline=pipe, draft_previous_tick=False)   # (engine calls are counted below)
    backend._generate_kwargs = lambda: {"use_cache": True, "num_beams": 1, "do_sample": False, "max_new_tokens": 24, "language": "en"}
    lens = [480000, 336000, 160000, 475679, 240000, 480000]
    bufs = [(wo.synth_audio(n, 40 + i, ["speechlike", "noise", "sine", "speechlike", "zeros", "noise"][i]), 1.5 * i, 16000)
            for i, n in enumerate(lens)]
    eng = pipe.model.engine
    single, passes_single = [], []
    for a, t0, sr in bufs:
        n0 = eng.calls["generate"]
        single.append(backend.transcribe(a.copy(), t0, sr))
        passes_single.append(eng.calls["generate"] - n0)
    assert max(passes_single) > 1, passes_single
    hub = BatchingHub(backend, max_batch=4, max_wait_s=0.3)
    done_at = {}
    futs = [hub.submit(a.copy(), t0, sr) for a, t0, sr in bufs[:2]]
    t_end = time.monotonic() + 120
    while hub.passes < 1 and time.monotonic() < t_end:     # the first two are one pass ahead ...
        time.sleep(0.01)
    futs += [hub.submit(a.copy(), t0, sr) for a, t0, sr in bufs[2:]]   # ... when four more arrive: passes mix seek positions
    for i, f in enumerate(futs):
        f.add_done_callback(lambda _f, i=i: done_at.setdefault(i, time.monotonic()))
    got = [f.result(timeout=600) for f in futs]
    assert hub._codec is not None and hub.passes > 0, "the continuous scheduler should have been used"
    batches = list(hub.batches)
    hub.close()
    assert normalise(got) == normalise(single)
    assert len(batches) == hub.passes < sum(passes_single), (batches, passes_single)
    assert sum(batches) == sum(passes_single) and max(batches) == 4, batches   # every chunk-pass ran exactly once, rows shared
    assert max(done_at[0], done_at[1]) <= min(done_at[i] for i in range(2, 6))  # answered when THEIR chunks were done


def test_hub_prefetches_the_encoder_stage_of_arrivals_during_a_pass():
    """`prefetch_cus > 0`: requests that arrive WHILE a pass decodes are opened and encoded on a sibling context by the prefetch
    thread, and the next pass adopts those rows (tw_adopt_cross_kv; here the CPU stand-in's sibling / adopt_cross_kv): results
    are exactly the per-request results, every chunk-pass still runs exactly once, and rows did take the prefetch route."""
    import time

    from tests.test_shortform import build
    from thewhisper_amd import AMDWhisperBackend
    from thewhisper_amd.serving import BatchingHub

    pipe = build(batch_size=4, chunk_s=30)
    backend = AMDWhisperBackend(None, chunk_length_s=30, asr_pipeline=pipe, draft_previous_tick=False)   # (engine calls are counted below)
    backend._generate_kwargs = lambda: {"use_cache": True, "num_beams": 1, "do_sample": False, "max_new_tokens": 24, "language": "en"}
    lens = [480000, 336000, 160000, 475679, 240000, 480000, 400000]
    bufs = [(wo.synth_audio(n, 40 + i, ["speechlike", "noise", "sine", "speechlike", "zeros", "noise", "sine"][i]), 1.5 * i, 16000)
            for i, n in enumerate(lens)]
    eng = pipe.model.engine
    single, passes_single = [], []
    for a, t0, sr in bufs:
        n0 = eng.calls["generate"]
        single.append(backend.transcribe(a.copy(), t0, sr))
        passes_single.append(eng.calls["generate"] - n0)
    hub = BatchingHub(backend, max_batch=4, max_wait_s=0.05, prefetch_cus=8)
    futs = [hub.submit(a.copy(), t0, sr) for a, t0, sr in bufs[:2]]
    t_end = time.monotonic() + 120
    while hub.passes < 1 and time.monotonic() < t_end:      # the first pass is decoding (seconds on the numpy engine) ...
        time.sleep(0.005)
    futs += [hub.submit(a.copy(), t0, sr) for a, t0, sr in bufs[2:5]]   # ... when three more arrive: the prefetch thread takes them
    while hub.passes < 3 and time.monotonic() < t_end:
        time.sleep(0.005)
    futs += [hub.submit(a.copy(), t0, sr) for a, t0, sr in bufs[5:]]
    got = [f.result(timeout=600) for f in futs]
    batches, prefetched = list(hub.batches), hub.prefetched
    pf = hub._prefetcher
    hub.close()
    assert pf is not None and not pf.thread.is_alive()
    assert normalise(got) == normalise(single)
    assert sum(batches) == sum(passes_single), (batches, passes_single)     # every chunk-pass ran exactly once
    assert prefetched >= 3, prefetched                                       # the late arrivals were encoded on the sibling


def test_hub_encodes_the_rows_that_sit_a_pass_out_on_the_side():
    """More requests in flight than a pass has rows, `prefetch_cus > 0`: the chunks that wait for the next pass - at whatever seek
    position they have reached - are encoded by the prefetch thread under the running pass's decode loop and adopted by the next
    pass (two cohorts taking turns).  Same results, every chunk-pass runs exactly once."""
    from tests.test_shortform import build
    from thewhisper_amd import AMDWhisperBackend
    from thewhisper_amd.serving import BatchingHub

    pipe = build(batch_size=4, chunk_s=30)
    backend = AMDWhisperBackend(None, chunk_length_s=30, asr_pipeline=pipe, draft_previous_tick=False)   # (engine calls are counted below)
    backend._generate_kwargs = lambda: {"use_cache": True, "num_beams": 1, "do_sample": False, "max_new_tokens": 24, "language": "en"}
    lens = [480000, 336000, 475679, 240000, 480000, 400000]
    bufs = [(wo.synth_audio(n, 60 + i, ["speechlike", "noise", "speechlike", "sine", "noise", "speechlike"][i]), 0.5 * i, 16000)
            for i, n in enumerate(lens)]
    eng = pipe.model.engine
    single, passes_single = [], []
    for a, t0, sr in bufs:
        n0 = eng.calls["generate"]
        single.append(backend.transcribe(a.copy(), t0, sr))
        passes_single.append(eng.calls["generate"] - n0)
    assert max(passes_single) >= 2, passes_single       # chunks that need further seek iterations: rows that CONTINUE
    hub = BatchingHub(backend, max_batch=2, max_wait_s=0.05, prefetch_cus=8)
    futs = [hub.submit(a.copy(), t0, sr) for a, t0, sr in bufs]     # six requests, two rows per pass
    got = [f.result(timeout=900) for f in futs]
    batches, ahead = list(hub.batches), hub.prefetched_ahead
    hub.close()
    assert normalise(got) == normalise(single)
    assert sum(batches) == sum(passes_single) and max(batches) == 2, (batches, passes_single)
    assert ahead >= 2, ahead            # rows with seek > 0 or of parked requests went through the side context


def test_hub_falls_back_to_whole_call_batches_when_the_call_is_not_eligible():
    from thewhisper_amd import AMDWhisperBackend
    from thewhisper_amd.serving import BatchingHub

    pipe = build_amd_pipeline("micro", 10, 2)
    pipe.model.fast_generate = False      # e.g. THEWHISPER_FAST_GENERATE=0: no plan can be learned
    backend = AMDWhisperBackend(None, chunk_length_s=10, asr_pipeline=pipe)
    hub = BatchingHub(backend, max_batch=2, max_wait_s=0.2)
    a = wo.synth_audio(64000, 3, "speechlike")
    want = backend.transcribe(a.copy(), 2.0, 16000)
    f1, f2 = hub.submit(a.copy(), 2.0, 16000), hub.submit(a.copy(), 2.0, 16000)
    assert normalise(f1.result(300)) == normalise(want) == normalise(f2.result(300))
    assert hub._codec is None and hub.passes == 1 and hub.rows == 2      # one whole-call batch carried both requests
    hub.close()


def test_continuous_hub_isolates_bad_requests_and_survives_an_engine_fault():
    from thewhisper_amd import AMDWhisperBackend
    from thewhisper_amd.serving import BatchingHub

    pipe = build_amd_pipeline("micro", 10, 2)
    backend = AMDWhisperBackend(None, chunk_length_s=10, asr_pipeline=pipe)
    good = wo.synth_audio(48000, 4, "speechlike")
    want = backend.transcribe(good.copy(), 0.0, 16000)
    hub = BatchingHub(backend, max_batch=2, max_wait_s=0.2)
    # a malformed buffer (not audio at all) fails its own future; its neighbour is served
    f_bad = hub.submit(np.array(["not audio"]), 0.0, 16000)
    f_ok = hub.submit(good.copy(), 0.0, 16000)
    assert normalise(f_ok.result(300)) == normalise(want)
    with pytest.raises(Exception):
        f_bad.result(300)
    assert hub._codec is not None and hub.passes >= 1
    # an engine fault fails exactly the requests that had a chunk in that pass; the hub keeps serving
    eng = pipe.model.engine
    inner, state = eng.generate_greedy, {"n": 0}

    def flaky(prompt, **kw):
        state["n"] += 1
        if state["n"] == 1:
            raise RuntimeError("injected device fault")
        return inner(prompt, **kw)

    eng.generate_greedy = flaky
    try:
        f1, f2 = hub.submit(good.copy(), 0.0, 16000), hub.submit(good.copy(), 1.0, 16000)
        for f in (f1, f2):
            with pytest.raises(RuntimeError, match="injected device fault"):
                f.result(300)
        f3 = hub.submit(good.copy(), 0.0, 16000)
        assert normalise(f3.result(300)) == normalise(want)
    finally:
        eng.generate_greedy = inner
        hub.close()


class _NoEngineBackend:
    """Just enough of a backend for a hub whose worker never sees a request (bookkeeping tests)."""

    class _P:
        class model:
            class engine:
                max_batch = 4

    asr_pipeline = _P()


def test_gather_window_decays_and_answer_table_is_pruned():
    """the turnaround estimate must not outlive the closed-loop callers that produced it, and the table of answered
    streams is pruned whether or not anybody gathers."""
    import time
    from concurrent.futures import Future

    from thewhisper_amd.serving import BatchingHub

    hub = BatchingHub(_NoEngineBackend(), max_batch=4, continuous=False)
    try:
        now = time.monotonic()
        with hub._lock:
            for i in range(8):                                     # a burst of closed-loop callers, 20-90 ms turnaround
                hub._turns.append((now - 0.5, 0.02 + 0.01 * i))
            hub._refresh_turnaround(now)
        assert 0.07 <= hub._turn_ema <= 0.091                     # 90th percentile of the recent samples
        hub._answered[1] = now
        hub._last_answer_t = now
        assert hub._gather_until(now) > now                        # a stream answered a moment ago: the intake stays open
        # ... ten seconds later nobody of that burst is around any more: no estimate, no gathering
        with hub._lock:
            hub._turns.clear()
            for i in range(8):
                hub._turns.append((now - 60.0, 0.05))
            hub._answered[2] = time.monotonic()
            hub._last_answer_t = time.monotonic()
        d = time.monotonic() + 0.001
        assert hub._gather_until(d) == d and hub._turn_ema == 0.0
        # real-time sessions only (_turn_ema stays 0): one entry per stream id, pruned by _answer itself
        old = time.monotonic() - 5.0
        with hub._lock:
            hub._answered = {1000 + k: old for k in range(300)}
        f = Future()
        f.stream_id = 7
        f.t_submit = time.monotonic()
        hub._answer(f, result=[])
        assert set(hub._answered) == {7}
    finally:
        hub.close()


def test_prefetcher_take_skips_rows_of_failed_requests():
    import collections
    import threading

    from thewhisper_amd.serving import _Prefetcher

    pf = _Prefetcher.__new__(_Prefetcher)
    pf.lock, pf.dead, pf.ahead, pf.count = threading.Lock(), [], [], 5
    w = [object() for _ in range(5)]
    pf.pre = collections.deque((w[i], i, f"seg{i}") for i in range(5))
    pf.ahead = [w[4], object()]
    gone = object()                              # a chunk of a failed request that is not with the prefetcher: nothing to remember
    pf.drop([w[0], w[2], w[4], gone])            # (round-5 advice: the work OBJECTS, not ids - ids of freed works are reused)
    assert len(pf.ahead) == 1 and len(pf.dead) == 3 and not any(x is gone for x in pf.dead)
    works, slot0, keep = pf.take(4)
    assert works == [w[1]] and slot0 == 1 and keep == ["seg1"]    # the run of consecutive slots ends in front of a dead row
    works, slot0, _ = pf.take(4)
    assert works == [w[3]] and slot0 == 3
    works, slot0, _ = pf.take(4)
    assert works == [] and pf.count == 0 and not pf.dead
