"""CPU: the full-depth golden vectors (oracle/make_golden_full.py - HF transformers fp32 at the true model sizes) are
self-consistent, and the numpy oracle reproduces them at full depth (large-v3, 32+32 layers, 10 s chunk)."""
import os

import numpy as np
import pytest

from oracle import whisper_oracle as wo
from tests.util import rel_l2

GOLD = os.path.join(os.path.dirname(__file__), "golden")
NAMES = ["full_large-v3_c10", "full_turbo_c30", "full_large-v3_c15", "full_large-v3_c10_b16", "full_large-v3_c20"]


@pytest.mark.parametrize("name", NAMES)
def test_full_depth_golden_is_self_consistent(name):
    z = np.load(os.path.join(GOLD, f"{name}.npz"))
    dims = wo.PRESETS[str(z["preset"])]
    seq, B, L = z["sequences"], z["sequences"].shape[0], z["sequences"].shape[1]
    assert (seq[:, :3] == [50258, 50259, 50360]).all() and L == 3 + int(z["max_new"])
    assert z["logits_top"].shape == (B, L, 8) and z["rand_logits_top"].shape[:2] == z["rand_ids"].shape
    assert (np.diff(z["logits_top"], axis=-1) <= 0).all()
    stride = int(z["logit_stride"])
    assert z["logits_sample"].shape[-1] == (dims.vocab + stride - 1) // stride
    # timestamp grammar: the first generated token is a timestamp <= max_initial_timestamp_index
    assert ((seq[:, 3] > 50364) & (seq[:, 3] <= 50365 + 50)).all()
    ts = z["token_timestamps"]
    assert ts.shape == seq.shape and (ts[:, :3] == 0).all() and (np.diff(ts[:, 3:], axis=1) >= -1e-6).all()
    assert ts.max() <= float(z["chunk_s"]) + 1e-6
    m = z["margins"][:, 2 : L - 1]
    assert np.isfinite(m).all() and (m >= 0).all()


@pytest.mark.parametrize("name", NAMES)
def test_stored_dtw_surface_reproduces_the_stored_timestamps(name):
    """`dtw_matrix` (what HF handed, negated, to its DTW) pins the word-timestamp stage at depth: the oracle's DTW on it gives
    exactly the stored token timestamps, and the margin helper of the GPU tests scores the reference's own jumps as optimal."""
    from tests.util import dtw_jump_margins

    z = np.load(os.path.join(GOLD, f"{name}.npz"))
    M, ts = z["dtw_matrix"], z["token_timestamps"]
    B, N, T = M.shape
    assert N == ts.shape[1] - 4 and T == 50 * int(z["chunk_s"])
    for b in range(min(B, 3)):
        ti, tj = wo.dtw(-M[b].astype(np.float64))
        jumps = np.pad(np.diff(ti), (1, 0), constant_values=1).astype(bool)
        jt = (tj[jumps] * 0.02).astype(np.float32)
        assert np.array_equal(jt, ts[b, 3:-1]) and ts[b, -1] == jt[-1]
    for b in range(B):
        mar, best = dtw_jump_margins(M[b], np.round(ts[b, 3:-1] / 0.02).astype(int))
        assert np.abs(mar).max() < 1e-9 and best < 0


def test_numpy_oracle_matches_full_depth_golden():
    """Pins the oracle to the reference arithmetic AT DEPTH: encoder rows and 6 teacher-forced decoder steps of the 32+32
    layer model (clip 0 of full_large-v3_c10).  float32 numpy vs float32 torch: agreement at accumulated round-off."""
    z = np.load(os.path.join(GOLD, "full_large-v3_c10.npz"))
    dims = wo.PRESETS["large-v3"]
    w = wo.make_weights(dims, int(z["weight_seed"]), scale=float(z["weight_scale"]), q_gain=float(z["q_gain"]))
    pcm = wo.synth_audio(160000, int(z["clip_seeds"][0]), str(z["clip_kinds"][0]))[None]
    mel = wo.log_mel(pcm, dims.n_mels)
    assert np.abs(mel[:, ::16, ::25] - z["mel_rows"][:1]).max() < 1e-4
    om = wo.OracleWhisper(dims, w, T=500)
    enc = om.encode(mel)
    assert rel_l2(enc[:, ::25, ::16], z["enc_rows"][:1]) < 1e-4
    ids = z["rand_ids"][:1, :6].astype(np.int64)
    lg, _ = om.decode(ids, om.new_cache(enc))
    assert rel_l2(lg[:, :, ::29], z["rand_logits_sample"][:1, :6]) < 1e-4
    top = np.take_along_axis(lg, z["rand_logits_top_idx"][:1, :6].astype(np.int64), axis=-1)
    assert np.abs(top - z["rand_logits_top"][:1, :6]).max() < 2e-3
