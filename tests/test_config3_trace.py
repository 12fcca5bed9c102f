"""BASELINE config 3's call pattern (tests/golden/config3_trace.json) IS what the reference's scheduler + stepper produce: it is
re-derived here from the reference package itself - /root/reference in the build container, the oracle/_ref bundle anywhere else
(oracle/ref_bundle.py) - and compared call by call.  bench.py's `config3` leg replays the committed trace on the MI355X."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from oracle import config3_trace as c3
from oracle import ref_bundle as rb

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden", "config3_trace.json")


def test_trace_shape_is_config3():
    g = json.load(open(GOLD))
    assert (g["seconds"], g["seed"], g["chunk_length_s"], g["min_process_chunk_s"], g["step_size_s"]) == (60, 0, 10, 0.5, 0.05)
    calls = g["calls"]
    # one backend call per 0.5 s of new audio once 2 s are buffered (R:...streaming_pipeline.py:759-760), to the end of the stream
    assert len(calls) == (60 - 2) * 2 + 1
    ends = [c["offset"] + c["n"] for c in calls]
    assert ends == [32000 + 8000 * i for i in range(len(calls))]
    # the rolling buffer never exceeds window - min_process (8.5 s) by more than one tick, and is trimmed, not reset
    assert max(c["n"] for c in calls) <= 9 * 16000 and min(c["n"] for c in calls) == 32000
    assert all(abs(c["offset"] / 16000 - c["t0"]) < 1e-3 for c in calls)
    starts = [c["offset"] for c in calls]
    assert starts == sorted(starts) and len(set(starts)) > 10        # trimmed a dozen times in a minute of steady speech


@pytest.mark.skipif(rb.reference_dir() is None, reason="neither /root/reference nor the oracle/_ref bundle is present")
def test_trace_is_what_the_reference_scheduler_produces():
    _, sp, streams = rb.import_reference()
    got = c3.trace(sp, streams)
    want = json.load(open(GOLD))
    assert got["calls"] == want["calls"] and got["committed_words"] == want["committed_words"]


def test_metronome_backend_is_consistent_across_overlapping_buffers():
    a = c3.MetronomeBackend.words(0.0, 9.0)
    b = c3.MetronomeBackend.words(4.03, 4.97)
    assert [w for w in a if w["start"] >= 4.03] == b                 # same words, same absolute times
    assert any(w["text"].endswith(".") for w in a) and any(w["text"].endswith(",") for w in a)
    assert all(w["end"] <= 9.0 - c3.TAIL_GUARD_S + 1e-9 for w in a)


@pytest.mark.skipif(not os.path.isdir(rb.REF), reason="the bundle is packed from /root/reference (build container only)")
def test_bundle_round_trip_without_the_checkout(tmp_path):
    """What the GPU box does: no /root/reference, only the archive -> the reference's classes import from a temporary directory."""
    out = rb.make_bundle(out=str(tmp_path / "pkg.tar.gz"))
    assert out and os.path.getsize(out) > 10000
    assert open(out, "rb").read() == open(rb.make_bundle(out=str(tmp_path / "again.tar.gz")), "rb").read()   # deterministic bytes
    code = (
        "import sys, json\n"
        "from oracle import ref_bundle as rb\n"
        f"rb.REF = '/nonexistent'; rb.BUNDLE = {out!r}\n"
        "d = rb.reference_dir(); assert d and d.startswith(__import__('tempfile').gettempdir()), d\n"
        "A, sp, st = rb.import_reference()\n"
        "assert sp.__file__.startswith(d) and hasattr(sp, 'StreamingPipeline') and hasattr(st, 'ArrayStream') and A.__name__ == 'ASRPipeline'\n"
        "print('OK', rb.which())\n")
    env = dict(os.environ)
    env.pop("TW_REFERENCE_DIR", None)
    r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "OK oracle/_ref bundle" in r.stdout, r.stderr[-2000:]
