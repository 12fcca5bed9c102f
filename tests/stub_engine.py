"""A GPU-less stand-in with the ``WhisperEngine`` surface bench.py drives (TW_BENCH_ENGINE="tests.stub_engine:make").
TESTS ONLY: it lets the N > 1 control flow of bench.py (rank discovery, barriers, max / sum / gather over ranks, the
one-JSON-line contract) run under gloo on a box without GPUs.  It computes nothing."""
import time

import numpy as np
import torch


class StubEngine:
    T = 500
    max_batch = 16

    def __init__(self):
        self._steps = 0

    def load_state_dict(self, sd):
        pass

    def logmel(self, pcm, **kw):
        return torch.zeros((pcm.shape[0], 128, 1000))

    def encode(self, mel, **kw):
        return None

    def cross_kv(self, B):
        pass

    def generate_greedy(self, prompt, max_new_tokens=128, **kw):
        time.sleep(0.01)
        B, n0 = prompt.shape
        self._steps = n0 + max_new_tokens - 1
        return {"sequences": np.zeros((B, n0 + max_new_tokens), np.int64), "length": n0 + max_new_tokens}

    def token_timestamps(self, B, n_prompt, L, nf, *a):
        return np.zeros((B, L), np.float32)

    def last_timings(self):
        return {"logmel_ms": 0.1, "encode_ms": 1.0, "cross_kv_ms": 0.2, "greedy_ms": 10.0, "token_timestamps_ms": 0.1,
                "decode_steps": self._steps}

    def close(self):
        pass


def make():
    return StubEngine()
