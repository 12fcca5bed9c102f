"""SURVEY.md section 8f-3, decoder side (opt-in): `AMDWhisperBackend(reuse_committed_prefix=True)` hands the previous tick's tokens
to the greedy loop as forced output (engine: tw_greedy_opts::n_forced, a batched prefill) when the new rolling buffer extends the
old one (R:thestage_speechkit/streaming/streaming_pipeline.py:770-796 re-decodes the whole buffer every 0.5 s).

CPU: the numpy engine (its `begin_index` restates the forced-output semantics).  Properties:
  * EXACT where the premise holds: the same buffer again -> the forced prefix is what a fresh decode picks -> identical words;
  * on the reference scheduler's own call sequence (golden stream: ragged rolling buffers) the option is an APPROXIMATION whose
    word-level delta is measured (reported; bounded loosely) - and it must engage only on calls that extend their predecessor;
  * off by default; a buffer that starts elsewhere (trim) or is longer than a chunk is decoded afresh."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import whisper_oracle as wo
from tests.test_pipeline_glue import build_amd_pipeline, normalise

torch.set_grad_enabled(False)
GOLD = os.path.join(os.path.dirname(__file__), "golden", "pipeline_golden.json")


def backends(device="cpu", engine_factory="oracle"):
    from tests.oracle_engine import oracle_engine_factory
    from thewhisper_amd import AMDWhisperBackend

    ef = oracle_engine_factory if engine_factory == "oracle" else None
    plain = AMDWhisperBackend(None, chunk_length_s=10, asr_pipeline=build_amd_pipeline("micro", 10, 1, device=device, engine_factory=ef),
                              draft_previous_tick=False)
    reuse = AMDWhisperBackend(None, chunk_length_s=10, asr_pipeline=build_amd_pipeline("micro", 10, 1, device=device, engine_factory=ef),
                              reuse_committed_prefix=True, reuse_margin_s=1.0)
    return plain, reuse


def word_delta(a, b):
    """(fraction of words of `a` found in `b` with the same text at the same index, max |start| difference over those)."""
    n = min(len(a), len(b))
    same = [i for i in range(n) if a[i]["text"] == b[i]["text"]]
    frac = len(same) / max(1, max(len(a), len(b)))
    dev = max([abs(a[i]["start"] - b[i]["start"]) for i in same], default=0.0)
    return frac, dev


def check_same_buffer_is_exact(plain, reuse):
    audio = wo.synth_audio(16000 * 6, 7, "speechlike")
    want = plain.transcribe(audio.copy(), 3.0, 16000)
    first = reuse.transcribe(audio.copy(), 3.0, 16000)
    again = reuse.transcribe(audio.copy(), 3.0, 16000)       # extends its predecessor trivially: the prefix is forced
    assert normalise(first) == normalise(want)
    assert normalise(again) == normalise(want)
    st = reuse.reuse_stats
    assert st["calls"] == 2 and st["reused"] == 1 and st["forced_tokens"] >= 2
    # elsewhere in the stream (another start time): decoded afresh
    other = reuse.transcribe(audio.copy(), 4.0, 16000)
    assert reuse.reuse_stats["reused"] == 1
    assert [w["text"] for w in other] == [w["text"] for w in want]
    # same start time and length but OTHER audio (a second session on a shared backend, a restarted stream): nothing is forced,
    # the result is the plain backend's
    other_audio = wo.synth_audio(16000 * 6, 8, "speechlike")
    reuse.transcribe(other_audio.copy(), 4.0, 16000)
    assert reuse.reuse_stats["reused"] == 1
    again2 = reuse.transcribe(audio.copy(), 4.0, 16000)
    assert reuse.reuse_stats["reused"] == 1 and [w["text"] for w in again2] == [w["text"] for w in want]
    # reset(): what a scheduler's clear() calls - the same buffer again is decoded afresh
    reuse.reset()
    reuse.transcribe(audio.copy(), 4.0, 16000)
    assert reuse.reuse_stats["reused"] == 1
    reuse.transcribe(audio.copy(), 4.0, 16000)
    assert reuse.reuse_stats["reused"] == 2


def test_off_by_default_and_exact_when_the_premise_holds():
    plain, reuse = backends()
    assert plain.reuse_committed_prefix is False
    check_same_buffer_is_exact(plain, reuse)


def replay_golden(plain, reuse, stride=1):
    with open(GOLD) as f:
        g = json.load(f)["streaming_micro_c10"]
    audio = wo.synth_audio(16000 * g["seconds"], g["seed"], g["kind"])
    fracs, devs, n_ext = [], [], 0
    prev = None
    for c in g["calls"][::stride]:
        buf = audio[c["offset"] : c["offset"] + c["n"]]
        a = plain.transcribe(buf.copy(), c["t0"], 16000)
        r0 = reuse.reuse_stats["reused"]
        b = reuse.transcribe(buf.copy(), c["t0"], 16000)
        engaged = reuse.reuse_stats["reused"] > r0
        extends = prev is not None and abs(prev["t0"] - c["t0"]) < 1e-6 and c["n"] >= prev["n"] and c["offset"] == prev["offset"]
        assert not engaged or extends, (c, prev)            # forced tokens only when the buffer extends its predecessor
        n_ext += int(extends)
        if a or b:
            f, d = word_delta(a, b)
            fracs.append(f)
            devs.append(d)
        for w in b:
            assert w["end"] >= w["start"] >= c["t0"] - 1e-6
        prev = c
    return fracs, devs, n_ext


def test_scheduler_call_sequence_with_reuse_cpu():
    plain, reuse = backends()
    fracs, devs, n_ext = replay_golden(plain, reuse, stride=1)
    st = reuse.reuse_stats
    print(f"\nREUSE (cpu stand-in): {st['reused']} of {st['calls']} calls reused a prefix ({n_ext} extend their predecessor), "
          f"{st['forced_tokens']} tokens forced / {st['decoded_tokens']} decoded; words identical to the plain backend: "
          f"mean {np.mean(fracs):.3f}, min {np.min(fracs):.3f}; max start deviation of identical words {np.max(devs):.2f} s")
    assert st["reused"] >= 1 and st["forced_tokens"] > 0
    assert np.mean(fracs) > 0.5        # an approximation, but not a different transcript


@pytest.mark.gpu
def test_reuse_on_the_mi355x():
    if not torch.cuda.is_available():
        pytest.skip("needs the MI355X")
    plain, reuse = backends(device="cuda", engine_factory=None)
    check_same_buffer_is_exact(plain, reuse)
    fracs, devs, n_ext = replay_golden(plain, reuse)
    st = reuse.reuse_stats
    print(f"\nREUSE (MI355X, micro model): {st['reused']} of {st['calls']} calls reused a prefix, {st['forced_tokens']} tokens forced / "
          f"{st['decoded_tokens']} decoded; words identical to the plain backend: mean {np.mean(fracs):.3f}, min {np.min(fracs):.3f}")
    assert st["reused"] >= 1 and np.mean(fracs) > 0.5


# ---- round 6: the exact form (tw_greedy_opts::n_draft) ---------------------------------------------------------------------------
def draft_backends(device="cpu", engine_factory="oracle"):
    from tests.oracle_engine import oracle_engine_factory
    from thewhisper_amd import AMDWhisperBackend

    ef = oracle_engine_factory if engine_factory == "oracle" else None
    plain = AMDWhisperBackend(None, chunk_length_s=10, asr_pipeline=build_amd_pipeline("micro", 10, 1, device=device, engine_factory=ef),
                              draft_previous_tick=False)
    draft = AMDWhisperBackend(None, chunk_length_s=10, asr_pipeline=build_amd_pipeline("micro", 10, 1, device=device, engine_factory=ef),
                              draft_previous_tick=True)
    return plain, draft


def test_draft_previous_tick_host_logic_cpu():
    """Host side of `draft_previous_tick` on the stand-in engine (whose `n_draft` is the definition: the call's result without the
    guesses): every call of the reference scheduler's sequence returns the plain backend's words; a draft is offered exactly on the
    calls that extend their predecessor, it is everything the predecessor produced, and the two options exclude each other."""
    from thewhisper_amd import AMDWhisperBackend

    plain, draft = draft_backends()
    with open(GOLD) as f:
        g = json.load(f)["streaming_micro_c10"]
    audio = wo.synth_audio(16000 * g["seconds"], g["seed"], g["kind"])
    prev, n_ext = None, 0
    for c in g["calls"]:
        buf = audio[c["offset"] : c["offset"] + c["n"]]
        a = plain.transcribe(buf.copy(), c["t0"], 16000)
        r0, d0 = draft.reuse_stats["reused"], draft.reuse_stats["draft_tokens"]
        last_ids = None if draft._last is None else draft._last["ids"]
        b = draft.transcribe(buf.copy(), c["t0"], 16000)
        assert a == b, c
        engaged = draft.reuse_stats["reused"] > r0
        extends = prev is not None and abs(prev["t0"] - c["t0"]) < 1e-6 and c["n"] >= prev["n"] and c["offset"] == prev["offset"]
        assert not engaged or extends, (c, prev)
        if engaged:
            assert draft.reuse_stats["draft_tokens"] - d0 == min(len(last_ids), 126)
        n_ext += int(extends)
        prev = c
    st = draft.reuse_stats
    print(f"\nDRAFT (cpu stand-in): {st['reused']} of {st['calls']} calls offered a draft ({n_ext} extend their predecessor), "
          f"{st['draft_tokens']} tokens offered, {st['confirmed_tokens']} confirmed")
    assert st["reused"] >= 1 and 0 < st["confirmed_tokens"] <= st["draft_tokens"] and st["forced_tokens"] == 0
    with pytest.raises(ValueError, match="alternatives"):
        AMDWhisperBackend(None, chunk_length_s=10, asr_pipeline=plain.asr_pipeline, reuse_committed_prefix=True, draft_previous_tick=True)
