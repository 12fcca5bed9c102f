"""Shared helpers for the parity tests (tests may import oracle/; the product package never does)."""
from __future__ import annotations

import dataclasses
from typing import Dict, Sequence, Tuple

import numpy as np

from oracle import whisper_oracle as wo


def dims_variant(name: str, **over) -> wo.WhisperDims:
    return dataclasses.replace(wo.PRESETS[name], **over)


def rel_l2(a: np.ndarray, b: np.ndarray) -> float:
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


def make_engine(dims: wo.WhisperDims, weights: Dict[str, np.ndarray], T: int, max_batch: int, dtype: str,
                heads: Sequence[Tuple[int, int]] = (), use_graph: bool = False):
    from thewhisper_amd.engine import WhisperEngine

    return WhisperEngine.from_numpy_weights(dataclasses.asdict(dims), weights, T=T, max_batch=max_batch, dtype=dtype,
                                            alignment_heads=list(heads), use_graph=use_graph)


KINDS = ["speechlike", "noise", "sine", "speechlike"]


def clips(n_samples: int, kinds) -> np.ndarray:
    """`kinds`: list of clip kinds, or an int B (cycles through KINDS with seed = index)."""
    if isinstance(kinds, int):
        kinds = [KINDS[i % len(KINDS)] for i in range(kinds)]
    return np.stack([wo.synth_audio(n_samples, seed=i, kind=k) for i, k in enumerate(kinds)])


PROMPT = [50258, 50259, 50360]
