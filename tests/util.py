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


# --------------------------------------------------------------------------------------------------------------------
# DTW margin rule (word timestamps in reduced precision).  The reference's alignment is the arg-min path of a dynamic
# programme over the cost surface C = -matrix (HF:models/whisper/generation_whisper.py:64-115): where two paths cost the same
# to within the arithmetic error of the surface, WHICH of them is returned is not a property of the audio.  For a token whose
# timestamp differs from the reference's, `dtw_jump_margins` measures - on the REFERENCE surface - how much more the best
# path that puts the token's jump where the engine put it costs than the reference's optimal path.  margin ~ 0 = a tie.
# --------------------------------------------------------------------------------------------------------------------
def _minplus_rows(C: np.ndarray) -> np.ndarray:
    """F[i, j] = cost of the cheapest monotone path (0,0) -> (i,j), both ends included; moves (+1,+1), (+1,0), (0,+1)."""
    n, m = C.shape
    F = np.empty((n, m), dtype=np.float64)
    F[0] = np.cumsum(C[0])
    for i in range(1, n):
        a = F[i - 1].copy()
        a[1:] = np.minimum(a[1:], F[i - 1, :-1])            # best predecessor in the row above: vertical or diagonal
        S = np.cumsum(C[i])
        Sprev = np.concatenate([[0.0], S[:-1]])
        F[i] = S + np.minimum.accumulate(a - Sprev)          # enter the row at some k <= j, then walk right
    return F


def dtw_jump_margins(matrix: np.ndarray, jump_frames: Sequence[int]) -> Tuple[np.ndarray, float]:
    """matrix: the reference's [N tokens, M frames] alignment matrix (the DTW runs on -matrix); jump_frames[i] = the frame at
    which token i starts in the path under test.  Returns (margins [N], optimal cost): margins[i] = cost of the cheapest
    path that enters row i at column jump_frames[i], minus the optimal cost (>= 0; 0 for the reference's own jumps)."""
    C = -np.asarray(matrix, dtype=np.float64)
    n, m = C.shape
    F = _minplus_rows(C)
    Bk = _minplus_rows(C[::-1, ::-1])[::-1, ::-1]           # cheapest path (i,j) -> (n-1,m-1), both ends included
    best = float(F[-1, -1])
    out = np.zeros(n, dtype=np.float64)
    for i in range(n):
        j = int(jump_frames[i])
        if i == 0:
            through = (F[0, j] + Bk[0, j] - C[0, j]) if j == 0 else np.inf   # the path starts in cell (0, 0)
        else:
            pred = F[i - 1, j] if j == 0 else min(F[i - 1, j], F[i - 1, j - 1])
            through = pred + Bk[i, j]
        out[i] = through - best
    return out, best


def alignment_matrix(cross: np.ndarray, num_input_ids: int, n_cols: int = None, median_width: int = 7) -> np.ndarray:
    """The [N, T] matrix the DTW runs on (negated), from raw alignment-head softmax rows [Ha, N_rows, T] of ONE stream:
    same steps as oracle.whisper_oracle.token_timestamps (HF:models/whisper/generation_whisper.py:333-360)."""
    w = np.asarray(cross, dtype=np.float32)
    if n_cols is not None:
        w = w[..., :n_cols]
    w = w[:, num_input_ids:, :]
    w = (w - w.mean(axis=-2, keepdims=True)) / w.std(axis=-2, keepdims=True)
    return wo.median_filter(w, median_width).mean(axis=0)
