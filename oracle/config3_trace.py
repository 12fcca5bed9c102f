"""BASELINE config 3's CALL PATTERN, produced by the reference's own scheduler (TEST INFRASTRUCTURE, see oracle/whisper_oracle.py).

SURVEY.md section 8d, config 3: a 60 s stream, seed 0, fed by `ArrayStream(step_size_s=0.05, real_time=False)`
(R:thestage_speechkit/streaming/streams.py:16-81) into `StreamingPipeline(chunk_length_s=10, min_process_chunk_s=0.5,
use_vad=False)` (R:thestage_speechkit/streaming/streaming_pipeline.py:443-531, :624-638, :740-822).  What the backend is asked
for - which rolling buffer, when - is decided by that scheduler from the words the backend returns (it trims the buffer at
sentence ends, commas and pauses, :854-935).  With random weights a large-v3 backend returns nothing (the reference's gibberish
filter, :41-43, :412-413), so the pattern a TRAINED model produces is generated here with a stand-in backend that answers like one:
a steady speaker, one word every 0.37 s on an absolute time grid (so overlapping buffers agree on their words, as a real
transcription does), a comma after every 5th and a full stop after every 11th word, no word closer than 0.1 s to the buffer's end.

`trace()` runs the REFERENCE scheduler and stepper around that stand-in and returns, per backend call, `(offset, n, t0)` - where
the buffer starts in the stream (samples), its length (samples), and the `buffer_start_time` the scheduler passed - no audio, no
text.  `oracle/make_golden.py` commits it as `tests/golden/config3_trace.json`; `bench.py` replays it on the MI355X at large-v3
dimensions (`config3` leg), `tests/test_config3_trace.py` re-derives it from the reference whenever the reference is importable.
"""
from __future__ import annotations

from typing import Any, Dict, List

import numpy as np

WORD_PERIOD_S = 0.37
WORD_ON_S = 0.05
WORD_OFF_S = 0.33
TAIL_GUARD_S = 0.1


class MetronomeBackend:
    """Duck-typed `TranscriptionBackend` (R:...streaming_pipeline.py:51-64): words on an absolute time grid."""

    def __init__(self):
        self.calls: List[Dict[str, Any]] = []

    @staticmethod
    def words(t0: float, duration: float) -> List[Dict[str, Any]]:
        out = []
        k = max(0, int(np.floor((t0 - WORD_ON_S) / WORD_PERIOD_S)))
        while True:
            start, end = k * WORD_PERIOD_S + WORD_ON_S, k * WORD_PERIOD_S + WORD_OFF_S
            if end > t0 + duration - TAIL_GUARD_S:
                break
            if start >= t0:
                text = f" w{k}"
                if k % 11 == 10:
                    text += "."
                elif k % 5 == 4:
                    text += ","
                out.append({"text": text, "start": round(start, 2), "end": round(end, 2)})
            k += 1
        return out

    def transcribe(self, audio: np.ndarray, buffer_start_time: float, sample_rate: int) -> List[Dict[str, Any]]:
        self.calls.append({"n": int(len(audio)), "t0": float(buffer_start_time), "first": float(audio[0]) if len(audio) else 0.0})
        return self.words(float(buffer_start_time), len(audio) / sample_rate)


def stream_audio(seconds: int = 60, seed: int = 0) -> np.ndarray:
    """`default_rng(seed).standard_normal * 0.1`, float32, clipped to [-1, 1]."""
    return (np.random.default_rng(seed).standard_normal(16000 * seconds) * 0.1).clip(-1, 1).astype(np.float32)


def trace(sp_module, streams_module, seconds: int = 60, seed: int = 0, chunk_length_s: int = 10,
          min_process_chunk_s: float = 0.5, step_size_s: float = 0.05) -> Dict[str, Any]:
    """Drive the reference's StreamingPipeline with its ArrayStream; `sp_module` / `streams_module` are the reference's
    `streaming_pipeline` and `streams` modules (oracle.ref_bundle.import_reference)."""
    audio = stream_audio(seconds, seed)
    backend = MetronomeBackend()
    pipe = sp_module.StreamingPipeline(backend=backend, chunk_length_s=chunk_length_s, min_process_chunk_s=min_process_chunk_s,
                                       use_vad=False)
    src = streams_module.ArrayStream(audio, step_size_s=step_size_s, sample_rate=16000, real_time=False)
    fed, n_committed, seen = 0, 0, 0
    calls: List[Dict[str, Any]] = []
    while True:
        chunk = src.next_chunk()
        if chunk is None:
            break
        fed += len(chunk)
        committed, _uncommitted = pipe(chunk)
        n_committed += len(committed)
        for c in backend.calls[seen:]:
            # the buffer handed to the backend ends with the newest sample the scheduler has taken in
            off = fed - c["n"]
            assert off >= 0 and float(audio[off]) == c["first"], "the buffer is not the newest c['n'] samples of the stream"
            calls.append({"offset": int(off), "n": c["n"], "t0": round(c["t0"], 6)})
        seen = len(backend.calls)
    return {
        "definition": "SURVEY.md section 8d config 3: reference StreamingPipeline(chunk_length_s=10, min_process_chunk_s=0.5, use_vad=False) "
                      "fed by the reference ArrayStream(step_size_s=0.05, real_time=False); stand-in backend = oracle/config3_trace.py::MetronomeBackend",
        "seconds": seconds, "seed": seed, "chunk_length_s": chunk_length_s, "min_process_chunk_s": min_process_chunk_s,
        "step_size_s": step_size_s, "calls": calls, "committed_words": n_committed,
    }
