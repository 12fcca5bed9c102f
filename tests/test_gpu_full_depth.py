"""Parity at BASELINE.json's REAL configurations (full depth): the HIP engine through the C ABI against golden vectors the
reference arithmetic produced on CPU (HF transformers fp32 behind R:thestage_speechkit/nvidia/asr_pipeline.py:57-60, run
by oracle/make_golden_full.py at the true model sizes with the oracle's seeded weights):

  full_large-v3_c10   large-v3 32+32 layers, T = 500   (configs 3/4)      bf16 + strict f32
  full_turbo_c30      turbo 32+4 layers,    T = 1500   (config 2)         bf16 (+ f32: two-pass 1500-key cross attention)
  full_large-v3_c15   large-v3 32+32 layers, T = 750   (config 5)         bf16 + fp8-vs-bf16 statement

Tolerances (stated once; measured values are printed by the tests and recorded in DESIGN.md section 2):
  log-mel            max-abs 2e-4 against the HF feature extractor rows
  strict f32         encoder rows rel-L2 <= 2e-4, teacher-forced logits rel-L2 <= 2e-4 per step and top-8 values within 2e-3,
                     greedy ids IDENTICAL wherever the golden margin exceeds 4 x 2e-3 (and the first step below that margin is
                     reported), token timestamps within one 0.02 s frame given identical ids
  bf16               encoder rows rel-L2 <= 3e-2, logits rel-L2 <= 5e-2 per step; top-1 identical on every step whose golden
                     raw top1-top2 margin exceeds 4 x the measured max top-8 abs error bound (0.25); greedy ids identical up to
                     the first step whose golden decision margin is below that bound (that step is reported)
The rel-L2 of a logits row is estimated on the stored stride-29 sample of the row (1789 of 51866 values).
"""
import os

import numpy as np
import pytest
import torch

from oracle import whisper_oracle as wo
from tests.util import make_engine, rel_l2

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")
PROMPT = [50258, 50259, 50360]
STRIDE = 29

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


def run_case(name, dtype, logit_tol, enc_tol, top_abs, check_ids=True):
    """Shared body: returns a dict of measured deviations (printed, asserted against the stated bounds)."""
    z, dims, w, pcm, heads = load_case(name)
    T, B = 50 * int(z["chunk_s"]), pcm.shape[0]
    eng = make_engine(dims, w, T=T, max_batch=B, dtype=dtype, heads=heads, use_graph=True)
    rep = {}
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
                worst_rel = max(worst_rel, rel_l2(lg[:, ::STRIDE], sample[:, s]))
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
            if same:
                ts = eng.token_timestamps(B, 3, got.shape[1], [2 * T] * B)
                dev = float(np.abs(ts[same] - z["token_timestamps"][same]).max())
                rep["token_ts_maxdev_s"] = dev
                rep["token_ts_exact_frac"] = float((np.abs(ts[same] - z["token_timestamps"][same]) < 1e-6).mean())
                rep["token_ts_within_1_frame_frac"] = float((np.abs(ts[same] - z["token_timestamps"][same]) <= 0.0201).mean())
                if dtype == "f32":
                    assert dev <= 0.0201, rep
                else:
                    # bf16 attention weights on the repetitive tail of a random-weight model's greedy path make the DTW
                    # path ill-conditioned (near-tied costs): most tokens stay within one frame, a few jump - reported
                    assert rep["token_ts_within_1_frame_frac"] >= 0.75, rep
    finally:
        eng.close()
    print(f"\nFULLDEPTH {name} {dtype}: " + ", ".join(f"{k}={v}" for k, v in rep.items()))
    return rep


# bounds: (logits rel-L2, encoder rel-L2, top-8 abs)
F32 = dict(logit_tol=2e-4, enc_tol=2e-4, top_abs=2e-3)
BF16 = dict(logit_tol=5e-2, enc_tol=3e-2, top_abs=0.25)


# ordered so that consecutive cases share the (6 GB, ~20 s to generate) seeded state dict
CASES = [("full_turbo_c30", "bf16"), ("full_turbo_c30", "f32"), ("full_large-v3_c10", "bf16"), ("full_large-v3_c10", "f32"),
         ("full_large-v3_c15", "bf16")]


@pytest.mark.parametrize("name,dtype", CASES)
def test_full_depth(name, dtype):
    rep = run_case(name, dtype, **(F32 if dtype == "f32" else BF16))
    if dtype == "f32" and rep["min_golden_margin"] > 4 * F32["top_abs"]:
        # strict mode: every decision margin on these clips is above the bound, so the ids must be identical outright
        assert all(d is None for d in rep["first_divergence(pos, golden_margin)"]), rep


def test_full_depth_fp8_vs_bf16_statement():
    """BASELINE config 5 (large-v3, 15 s chunks, MXFP8 decoder weights + fp8 cross-K/V caches): full-depth logit error of the fp8
    context against the fp32 reference, beside bf16's (no reference exists for fp8 results; this states the quantisation error
    at depth: measured 8.5e-2 vs 1.5e-2 for bf16).
    Bounds: rel-L2 <= 0.25 and top-1 identical wherever the golden margin exceeds 1.5."""
    name = "full_large-v3_c15"
    z, dims, w, pcm, heads = load_case(name)
    T, B = 50 * int(z["chunk_s"]), pcm.shape[0]
    eng = make_engine(dims, w, T=T, max_batch=B, dtype="fp8", heads=heads)
    try:
        eng.encode(eng.logmel(torch.from_numpy(pcm).cuda()))
        eng.cross_kv(B)
        eng.decoder_reset(B)
        ids = z["rand_ids"].astype(np.int64)
        worst = 0.0
        for s in range(ids.shape[1]):
            lg = eng.decode_step(ids[:, s].tolist()).cpu().numpy()
            worst = max(worst, rel_l2(lg[:, ::STRIDE], z["rand_logits_sample"][:, s]))
            for b in range(B):
                if z["rand_logits_top"][b, s, 0] - z["rand_logits_top"][b, s, 1] > 1.5:
                    assert int(lg[b].argmax()) == int(z["rand_logits_top_idx"][b, s, 0])
        print(f"\nFULLDEPTH {name} fp8: rand_path_logits_rel_l2={worst}")
        assert worst < 0.25
    finally:
        eng.close()
