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
