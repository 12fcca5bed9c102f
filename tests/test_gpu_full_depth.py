"""Parity at BASELINE.json's REAL configurations (full depth): the HIP engine through the C ABI against golden vectors the
reference arithmetic produced on CPU (HF transformers fp32 behind R:thestage_speechkit/nvidia/asr_pipeline.py:57-60, run
by oracle/make_golden_full.py at the true model sizes with the oracle's seeded weights):

  full_large-v3_c10   large-v3 32+32 layers, T = 500   (configs 3/4)      bf16 + strict f32
  full_turbo_c30      turbo 32+4 layers,    T = 1500   (config 2)         bf16 (+ f32: two-pass 1500-key cross attention)
  full_large-v3_c15   large-v3 32+32 layers, T = 750   (config 5)         bf16 + the fp8 context (ids, timestamps, logits)
  full_large-v3_c10_b16  large-v3, T = 500, 16 clips x 160 new tokens (what bench.py times)   bf16 + strict f32

Tolerances (stated once; measured values are printed by the tests and recorded in DESIGN.md section 2):
  log-mel            max-abs 2e-4 against the HF feature extractor rows
  strict f32         encoder rows rel-L2 <= 2e-4, teacher-forced logits rel-L2 <= 2e-4 per step and top-8 values within 2e-3,
                     greedy ids IDENTICAL wherever the golden margin exceeds 4 x 2e-3 (and the first step below that margin is
                     reported), token timestamps within one 0.02 s frame given identical ids
  bf16               encoder rows rel-L2 <= 3e-2, logits rel-L2 <= 3e-2 per step, top-8 logit values within 0.12; top-1 identical
                     on every step whose golden raw top1-top2 margin exceeds 4 x 0.12; greedy ids identical up to the first step
                     whose golden decision margin is below that bound (that step is reported)
  fp8 (config 5)     logits rel-L2 <= 0.13 (1.5 x the measured 0.087), top-8 within FP8["top_abs"], same margin rules
  token timestamps   (every dtype, given identical ids) identical +-0.02 s wherever the DTW margin exceeds EPS_DTW: a token may
                     sit elsewhere ONLY if, on the REFERENCE's own cost surface (`dtw_matrix` of the golden file), the best path
                     with the token's jump at the engine's frame costs within EPS_DTW of the optimal path (tests/util.py:
                     dtw_jump_margins) - i.e. the reference's arg-min was a tie at the surface's own resolution
The rel-L2 of a logits row is estimated on the stored strided sample of the row (stride 29: 1789 of 51866 values; 233 for
the 16-clip case).
"""
import os

import numpy as np
import pytest
import torch

from oracle import whisper_oracle as wo
from tests.util import alignment_matrix, dtw_jump_margins, make_engine, rel_l2

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")
PROMPT = [50258, 50259, 50360]
STRIDE = 29
# DTW margin (cost units of the z-scored alignment matrix, whose cells are O(1) and whose optimal paths cost ~ -250): see above
EPS_DTW = 0.05
DUMP = os.environ.get("TW_DUMP_DIR")   # diagnostics: engine ids / timestamps / alignment rows of every case as .npz

_weights_cache = {}


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("needs the MI355X")
    yield
    _weights_cache.clear()


def load_case(name):
    z = np.load(os.path.join(GOLD, f"{name}.npz"))
    dims = wo.PRESETS[str(z["preset"])]
    key = (str(z["preset"]), int(z["weight_seed"]), float(z["weight_scale"]), float(z["q_gain"]))
    if key not in _weights_cache:
        _weights_cache.clear()  # one full-size float32 state dict (6 GB) at a time
        _weights_cache[key] = wo.make_weights(dims, key[1], scale=key[2], q_gain=key[3])
    pcm = np.stack([wo.synth_audio(16000 * int(z["chunk_s"]), int(s), str(k)) for k, s in zip(z["clip_kinds"], z["clip_seeds"])])
    heads = [tuple(int(x) for x in h) for h in z["alignment_heads"]]
    return z, dims, _weights_cache[key], pcm, heads


def check_timestamps(z, ts, streams, dtype, dump=None):
    """Token timestamps of `streams` (engine ids == golden ids there) against the golden ones: identical +-one frame, except
    where the reference's own DTW was a tie (margin rule, module docstring).  Returns the measured statistics."""
    gts = z["token_timestamps"]
    dev = np.abs(ts[streams] - gts[streams])
    rep = {"token_ts_maxdev_s": float(dev.max()), "token_ts_exact_frac": float((dev < 1e-6).mean()),
           "token_ts_within_1_frame_frac": float((dev <= 0.0201).mean())}
    worst_margin, n_moved, margins_moved = 0.0, 0, []
    for b in streams:
        moved = np.nonzero(np.abs(ts[b, 3:-1] - gts[b, 3:-1]) > 0.0201)[0]
        if len(moved) == 0:
            continue
        jf = np.round(ts[b, 3:-1] / 0.02).astype(int)
        mar, _ = dtw_jump_margins(z["dtw_matrix"][b], jf)
        n_moved += len(moved)
        margins_moved.extend(float(x) for x in mar[moved])
        worst_margin = max(worst_margin, float(mar[moved].max()))
    rep["tokens_moved_gt_1_frame"] = n_moved
    rep["worst_dtw_margin_of_a_moved_token"] = worst_margin
    if dump is not None:
        dump["moved_margins"] = np.array(margins_moved)
    if dtype == "f32":
        assert dev.max() <= 0.0201, rep
    assert worst_margin <= EPS_DTW, rep
    # the last entry duplicates the last jump (HF :377-379)
    return rep


def run_case(name, dtype, logit_tol, enc_tol, top_abs, check_ids=True):
    """Shared body: returns a dict of measured deviations (printed, asserted against the stated bounds)."""
    z, dims, w, pcm, heads = load_case(name)
    T, B = 50 * int(z["chunk_s"]), pcm.shape[0]
    stride = int(z["logit_stride"]) if "logit_stride" in z.files else STRIDE
    eng = make_engine(dims, w, T=T, max_batch=B, dtype=dtype, heads=heads, use_graph=True)
    rep = {}
    dump = {}
    try:
        # A1
        mel = eng.logmel(torch.from_numpy(pcm).cuda(), out_dtype=torch.float32)
        rep["mel_maxabs"] = float(np.abs(mel.cpu().numpy()[:, ::16, ::25] - z["mel_rows"]).max())
        assert rep["mel_maxabs"] < 2e-4
        # A2-A4 at full depth
        enc = eng.encode(mel, return_hidden=True).cpu().numpy()
        assert np.isfinite(enc).all()
        rep["enc_rel_l2"] = rel_l2(enc[:, ::25, ::16], z["enc_rows"])
        rep["enc_norm_ratio"] = float(np.abs(np.linalg.norm(enc.reshape(B, -1), axis=1) / z["enc_norm"] - 1).max())
        assert rep["enc_rel_l2"] < enc_tol, rep
        eng.cross_kv(B)

        # A5-A8 teacher-forced along the reference's greedy path and over random tokens
        def teacher(ids, tops, top_idx, sample):
            eng.decoder_reset(B)
            worst_rel, worst_top, flips = 0.0, 0.0, []
            for s in range(ids.shape[1]):
                lg = eng.decode_step(ids[:, s].tolist()).cpu().numpy()
                worst_rel = max(worst_rel, rel_l2(lg[:, ::stride], sample[:, s]))
                for b in range(B):
                    worst_top = max(worst_top, float(np.abs(lg[b, top_idx[b, s]] - tops[b, s]).max()))
                    margin = tops[b, s, 0] - tops[b, s, 1]
                    if margin > 4 * top_abs:
                        assert int(lg[b].argmax()) == int(top_idx[b, s, 0]), (name, dtype, b, s, margin)
                    elif int(lg[b].argmax()) != int(top_idx[b, s, 0]):
                        flips.append((b, s, float(margin)))
            return worst_rel, worst_top, flips

        seq = z["sequences"].astype(np.int64)
        rep["greedy_path_logits_rel_l2"], rep["greedy_path_top8_maxabs"], rep["greedy_path_subm_flips"] = teacher(
            seq[:, :-1], z["logits_top"], z["logits_top_idx"], z["logits_sample"])
        # A11 given IDENTICAL ids by construction: the teacher-forced pass left the alignment heads' softmax rows of the
        # reference's own greedy path in the context, for every stream (whether or not the free-running loop below stays on it)
        Lg = seq.shape[1]
        ts_tf = eng.token_timestamps(B, 3, Lg, [2 * T] * B)
        rep.update(check_timestamps(z, ts_tf, list(range(B)), dtype, dump if DUMP else None))
        if DUMP:
            al = eng.get_alignment(B, Lg - 1)
            dump.update(ts_teacher_forced=ts_tf, matrix_teacher_forced=np.stack([alignment_matrix(al[b], 3) for b in range(B)]))
            del al
        rep["rand_path_logits_rel_l2"], rep["rand_path_top8_maxabs"], rep["rand_path_subm_flips"] = teacher(
            z["rand_ids"].astype(np.int64), z["rand_logits_top"], z["rand_logits_top_idx"], z["rand_logits_sample"])
        assert rep["greedy_path_logits_rel_l2"] < logit_tol and rep["rand_path_logits_rel_l2"] < logit_tol, rep
        assert rep["greedy_path_top8_maxabs"] < top_abs and rep["rand_path_top8_maxabs"] < top_abs, rep

        # A9-A11 free-running greedy with the timestamp grammar + token timestamps
        if check_ids:
            prompt = np.tile(np.array(PROMPT, dtype=np.int32), (B, 1))
            out = eng.generate_greedy(prompt, max_new_tokens=int(z["max_new"]), timestamps=True, want_alignment=True)
            got = out["sequences"]
            L = min(got.shape[1], seq.shape[1])
            margins = z["margins"]
            first_div = []
            for b in range(B):
                neq = np.nonzero(got[b, :L] != seq[b, :L])[0]
                if len(neq) == 0:
                    first_div.append(None)
                    continue
                p = int(neq[0])              # token at position p was decided at step p-1
                m = float(margins[b, p - 1])
                first_div.append((p, m))
                # identical up to the first sub-margin decision: a divergence is only legitimate there
                assert m <= 4 * top_abs, f"{name}/{dtype}: stream {b} diverges at position {p} where the golden margin is {m}"
            rep["first_divergence(pos, golden_margin)"] = first_div
            rep["min_golden_margin"] = float(margins[:, 2 : seq.shape[1] - 1].min())
            same = [b for b in range(B) if first_div[b] is None and got.shape[1] == seq.shape[1]]
            rep["streams_with_identical_ids"] = len(same)
            ts = eng.token_timestamps(B, 3, got.shape[1], [2 * T] * B)
            if DUMP:
                dump.update(ids=got, ts=ts)
            if same:
                # same ids, same arithmetic (captured step graph or not): the free-running loop's timestamps ARE the teacher-forced ones
                assert np.array_equal(ts[same], ts_tf[same]), (name, dtype, "free-running vs teacher-forced token timestamps")
    finally:
        eng.close()
        if DUMP and dump:
            os.makedirs(DUMP, exist_ok=True)
            np.savez_compressed(os.path.join(DUMP, f"{name}_{dtype}.npz"), **dump)
    print(f"\nFULLDEPTH {name} {dtype}: " + ", ".join(f"{k}={v}" for k, v in rep.items()))
    return rep


# bounds: (logits rel-L2, encoder rel-L2, top-8 abs)
F32 = dict(logit_tol=2e-4, enc_tol=2e-4, top_abs=2e-3)
BF16 = dict(logit_tol=3e-2, enc_tol=3e-2, top_abs=0.12)
# MXFP8 decoder weights + e4m3 cross-K/V (BASELINE config 5): the encoder is bf16, so its bound is bf16's
FP8 = dict(logit_tol=0.13, enc_tol=3e-2, top_abs=0.6)


# ordered so that consecutive cases share the (6 GB, ~20 s to generate) seeded state dict
CASES = [("full_turbo_c30", "bf16"), ("full_turbo_c30", "f32"), ("full_large-v3_c10", "bf16"), ("full_large-v3_c10", "f32"),
         ("full_large-v3_c10_b16", "bf16"), ("full_large-v3_c10_b16", "f32"), ("full_large-v3_c15", "bf16"),
         ("full_large-v3_c15", "fp8")]


@pytest.mark.parametrize("name,dtype", CASES)
def test_full_depth(name, dtype):
    rep = run_case(name, dtype, **{"f32": F32, "bf16": BF16, "fp8": FP8}[dtype])
    if dtype == "f32" and rep["min_golden_margin"] > 4 * F32["top_abs"]:
        # strict mode: every decision margin on these clips is above the bound, so the ids must be identical outright
        assert all(d is None for d in rep["first_divergence(pos, golden_margin)"]), rep
