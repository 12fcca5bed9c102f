"""TESTS ONLY: the ``host_factory`` the 2-process NodeRouter test hands to its workers - the micro model on the CPU stand-in
engine (tests/oracle_engine.py) behind a BatchingHub, with a minimal per-session scheduler (rolling buffer -> one backend call)."""
import numpy as np
import torch


class TinyScheduler:
    """Stand-in for the reference's StreamingPipeline where it is not importable: accumulate, then transcribe the buffer."""

    def __init__(self, backend, chunk_length_s):
        self.backend = backend
        self.buf = np.zeros(0, np.float32)
        self.max = int(chunk_length_s * 16000)

    def add_new_chunk(self, chunk):
        self.buf = np.concatenate([self.buf, np.asarray(chunk, np.float32)])[-self.max:]

    def process_new_chunk(self):
        if len(self.buf) < 8000:
            return [], []
        return [], self.backend.transcribe(self.buf, 0.0, 16000)

    def clear(self):
        self.buf = np.zeros(0, np.float32)


def make_host(rank, world, max_batch=4, **_):
    torch.set_grad_enabled(False)
    torch.set_num_threads(2)
    from tests.test_pipeline_glue import build_amd_pipeline
    from thewhisper_amd import AMDWhisperBackend
    from thewhisper_amd.gateway import SessionHost
    from thewhisper_amd.serving import BatchingHub

    pipe = build_amd_pipeline("micro", 10, max_batch)
    backend = AMDWhisperBackend(None, chunk_length_s=10, asr_pipeline=pipe)
    host = SessionHost(BatchingHub(backend, max_batch=max_batch, max_wait_s=0.2), scheduler_factory=TinyScheduler)
    host.rank = rank
    return host


class _StubBackend:
    """No model at all: a pass over up to ``max_batch`` requests takes ``pass_s`` seconds of wall time (what a 16-row pass of
    the real engine takes, profiles/r03_SUMMARY.md: ~200 ms) and answers one word per request.  Lets tests/test_node_scale.py
    drive the REAL front end - FastAPI app, NodeRouter, pipes, worker thread pools, SessionHost, BatchingHub - at BASELINE
    config 4's load (128 sessions on 8 ranks) on a machine without GPUs."""

    sample_rate = 16000
    chunk_length_s = 10

    def __init__(self, max_batch, pass_s):
        from types import SimpleNamespace

        self.pass_s = pass_s
        self.asr_pipeline = SimpleNamespace(model=SimpleNamespace(engine=SimpleNamespace(max_batch=max_batch)))

    def transcribe_many(self, requests, batch_size=None):
        import time

        time.sleep(self.pass_s)
        return [[{"text": f" n{len(a)}", "start": float(t0), "end": float(t0) + len(a) / sr}] for a, t0, sr in requests]

    def transcribe(self, audio, buffer_start_time, sample_rate):
        return self.transcribe_many([(audio, buffer_start_time, sample_rate)])[0]


def make_stub_host(rank, world, max_batch=16, pass_s=0.2, max_wait_s=0.02, **_):
    from thewhisper_amd.gateway import SessionHost
    from thewhisper_amd.serving import BatchingHub

    host = SessionHost(BatchingHub(_StubBackend(max_batch, pass_s), max_batch=max_batch, max_wait_s=max_wait_s),
                       scheduler_factory=TinyScheduler)
    host.rank = rank
    return host
