"""Parity at BASELINE.json's REAL configurations (full depth): the HIP engine through the C ABI against golden vectors the
reference arithmetic produced on CPU (HF transformers fp32 behind R:thestage_speechkit/nvidia/asr_pipeline.py:57-60, run
by oracle/make_golden_full.py at the true model sizes with the oracle's seeded weights):

  full_large-v3_c10   large-v3 32+32 layers, T = 500   (configs 3/4)      bf16 + strict f32
  full_turbo_c30      turbo 32+4 layers,    T = 1500   (config 2)         bf16 (+ f32: two-pass 1500-key cross attention)
  full_large-v3_c15   large-v3 32+32 layers, T = 750   (config 5)         bf16 + the fp8 context (ids, timestamps, logits)
  full_large-v3_c20   large-v3 32+32 layers, T = 1000  (20 s chunks)      bf16 + f16 + strict f32
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
  token timestamps   given identical ids (teacher-forced along the reference's greedy path, all streams): see check_timestamps -
                     exact stage parity on the engine's own alignment rows, a bound on the alignment surface, and strict f32
                     within one 0.02 s frame; in reduced precision a token may move only within what that surface error can
                     overturn on the reference's own cost surface (`dtw_matrix` of the golden file)
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
DUMP = os.environ.get("TW_DUMP_DIR")   # diagnostics: engine ids / timestamps / alignment rows of every case as .npz

_weights_cache = {}


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("needs the MI355X")
    yield
    _weights_cache.clear()


def load_ctrl(name):
    """Reduced-precision CONTROL of the case (oracle/make_golden_ctrl.py): the reference's own arithmetic (HF on CPU) cast to
    bf16 / fp16, teacher-forced along the same fp32 greedy path, its surface and token timestamps - the yardstick for what a
    bf16 engine may lose against the fp32 golden."""
    p = os.path.join(GOLD, f"ctrl_{name}.npz")
    return np.load(p) if os.path.exists(p) else None


def load_fp8(name):
    """The W8A16 context's EXTERNAL golden (oracle/make_golden_fp8.py): the reference's model class in float32 over exactly the
    parameters that context stores (decoder weights de-quantised from MXFP8 with torch's float8_e4m3fn, LayerNorms folded, cross-K/V
    through an fp8 round trip), teacher-forced along the fp32 greedy path."""
    p = os.path.join(GOLD, f"fp8_{name}.npz")
    return np.load(p) if os.path.exists(p) else None


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


def check_vs_control(z, ctrl, mats, ts, rep, problems, slack=1.25, tag="bf16"):
    """bf16 engine vs the reference's OWN bf16 arithmetic (HF `.to(torch.bfloat16)` on CPU, `ctrl`), both against the fp32
    golden, on the clips the control covers: the engine's alignment surface and its token timestamps may be no further from
    the fp32 reference than `slack` x what HF-bf16 itself is (timestamps: tokens outside one frame, + 2 tokens per clip of
    counting noise - the DTW arg-min moves in whole tokens).  HF-fp16 (the reference's streaming default,
    R:thestage_speechkit/streaming/streaming_pipeline.py:369-370) is reported beside it."""
    clips = [int(c) for c in ctrl["clips"]]
    gts, Mg = z["token_timestamps"], z["dtw_matrix"]
    e_rel = [rel_l2(mats[c], Mg[c]) for c in clips]
    h_rel = [rel_l2(ctrl[f"{tag}_dtw_matrix"][i], Mg[c]) for i, c in enumerate(clips)]
    f_rel = [rel_l2(ctrl["fp16_dtw_matrix"][i], Mg[c]) for i, c in enumerate(clips)]
    e_out = [int((np.abs(ts[c] - gts[c]) > 0.0201).sum()) for c in clips]
    h_out = [int((np.abs(ctrl[f"{tag}_token_timestamps"][i] - gts[c]) > 0.0201).sum()) for i, c in enumerate(clips)]
    f_out = [int((np.abs(ctrl["fp16_token_timestamps"][i] - gts[c]) > 0.0201).sum()) for i, c in enumerate(clips)]
    n_tok = gts.shape[1]
    rep.update(ctrl_clips=clips, surface_rel_l2_engine=[round(x, 4) for x in e_rel], surface_rel_l2_hf_bf16=[round(x, 4) for x in h_rel],
               surface_rel_l2_hf_fp16=[round(x, 4) for x in f_rel], tokens_outside_1_frame_engine=e_out,
               tokens_outside_1_frame_hf_bf16=h_out, tokens_outside_1_frame_hf_fp16=f_out, tokens_per_clip=n_tok,
               worst_dev_s_engine=float(max(np.abs(ts[c] - gts[c]).max() for c in clips)),
               worst_dev_s_hf_bf16=float(max(np.abs(ctrl[f"{tag}_token_timestamps"][i] - gts[c]).max() for i, c in enumerate(clips))),
               control=f"HF-{tag} (the figures labelled hf_bf16 are this control's)")
    # the DTW path's excess cost on the fp32 reference surface (fraction of the optimum), engine vs HF-bf16: reported
    def excess(M, Mg_):
        Cg = -Mg_.astype(np.float64)
        ti, tj = wo.dtw(-M.astype(np.float64))
        gi, gj = wo.dtw(Cg)
        return float(Cg[ti, tj].sum() - Cg[gi, gj].sum()) / abs(float(Cg[gi, gj].sum()))
    rep.update(path_excess_engine=[round(excess(mats[c], Mg[c]), 5) for c in clips],
               path_excess_hf_bf16=[round(excess(ctrl[f"{tag}_dtw_matrix"][i], Mg[c]), 5) for i, c in enumerate(clips)])
    for c, e, h in zip(clips, e_rel, h_rel):
        if e > slack * h:
            problems.append(f"clip {c}: engine alignment surface rel-L2 {e:.4f} > {slack} x HF-bf16's own {h:.4f}")
    # Tokens outside one frame are REPORTED, not asserted: the arg-min path over these random-weight surfaces is chaotic - one
    # bifurcation moves twenty tokens.  HF-bf16 ITSELF moves 5, 34, 46 and 92 of 163 tokens on the four control clips of the 16-clip
    # case and 110 / 45 / 9 / 83 of 131 on the 15 s clips (engine there: 58 / 47 / 9 / 56); two engine builds that differ by one fp32
    # rounding in the encoder softmax moved 191 and 255 tokens on the former (HF-bf16: 177); and on the first 35-token clip of the
    # 2-clip case the build with the encoder LayerNorms folded moves 21 (HF-bf16: 2) while being CLOSER to the fp32 surface than
    # HF-bf16 (0.105 vs 0.114).  A bound on that count fails or passes by the build's rounding, not by its quality.  What IS asserted
    # about the consequence: the engine's path is a near-optimal path of the REFERENCE surface - its excess cost there is within
    # the error budget that optimality on its own surface implies (exact inequality, check_timestamps 3a) and within a fraction of the
    # optimum proportional to the surface tolerance (3b) - and the gross alarm on the share of timestamps within one frame.
    rep["tokens_outside_1_frame_rule"] = "reported (chaotic arg-min); asserted: surface <= 1.25 x control, path excess bounds"


def check_timestamps(z, eng, ts, streams, dtype, n_rows, bounds, dump=None, ctrl=None):
    """Word-timestamp stage (A11) of `streams` (engine ids == golden ids there) against the reference, in three statements:
      1. STAGE PARITY, exact: the engine's timestamps are the reference algorithm (z-score, median filter, head mean, DTW with
         its tie-breaks: oracle restatement pinned to HF on CPU) applied to the engine's OWN alignment rows - bit for bit;
      2. SURFACE: the engine's alignment matrix is within `bounds` of the reference's `dtw_matrix` (rel-L2 and max-abs);
      3. CONSEQUENCE: strict f32 - every timestamp within one 0.02 s frame; reduced precision - a token may sit elsewhere only
         where the reference's arg-min is within reach of that surface error: the engine path's excess cost on the REFERENCE
         surface is (a) at most the sum of |surface error| over the cells where the two paths differ (what optimality on the
         engine's surface implies) and (b) at most `bounds["excess_frac"]` of the optimal cost; per moved token the margin
         (dtw_jump_margins) is reported.  On these random-weight fixtures the reference's own per-token margins are 0.1-0.3
         cost units at an optimal cost of ~ -250 (tests/test_golden_full.py), i.e. the surface is flat almost everywhere."""
    gts, Mg = z["token_timestamps"], z["dtw_matrix"]
    al = eng.get_alignment(len(gts), n_rows)
    dev = np.abs(ts[streams] - gts[streams])
    rep = {"token_ts_maxdev_s": float(dev.max()), "token_ts_exact_frac": float((dev < 1e-6).mean()),
           "token_ts_within_1_frame_frac": float((dev <= 0.0201).mean())}
    worst_margin, n_moved, worst_excess_frac, surf_rel, surf_abs, problems = 0.0, 0, 0.0, 0.0, 0.0, []
    mats = []
    for b in streams:
        Me = alignment_matrix(al[b], 3)
        mats.append(Me)
        surf_rel = max(surf_rel, rel_l2(Me, Mg[b]))
        surf_abs = max(surf_abs, float(np.abs(Me - Mg[b]).max()))
        # 1. the engine's timestamps = the reference algorithm on the engine's rows
        ti, tj = wo.dtw(-Me.astype(np.float64))
        jumps = np.pad(np.diff(ti), (1, 0), constant_values=1).astype(bool)
        jt = (tj[jumps] * 0.02).astype(np.float32)
        if not (np.array_equal(jt, ts[b, 3:-1]) and ts[b, -1] == jt[-1] and (ts[b, :3] == 0).all()):
            problems.append(f"stream {b}: engine timestamps are not the DTW of the engine's own alignment rows")
        # 3. the engine path on the reference surface
        Cg = -Mg[b].astype(np.float64)
        gi, gj = wo.dtw(Cg)
        on_e = np.zeros(Cg.shape, bool); on_e[ti, tj] = True
        on_g = np.zeros(Cg.shape, bool); on_g[gi, gj] = True
        excess = float(Cg[on_e].sum() - Cg[on_g].sum())
        budget = float(np.abs(Me.astype(np.float64) - Mg[b])[on_e ^ on_g].sum())
        if excess > budget + 1e-3:
            problems.append(f"stream {b}: excess cost {excess:.4f} on the reference surface exceeds the error budget {budget:.4f}")
        worst_excess_frac = max(worst_excess_frac, excess / abs(float(Cg[on_g].sum())))
        moved = np.nonzero(np.abs(ts[b, 3:-1] - gts[b, 3:-1]) > 0.0201)[0]
        if len(moved):
            mar, _ = dtw_jump_margins(Mg[b], np.round(ts[b, 3:-1] / 0.02).astype(int))
            n_moved += len(moved)
            worst_margin = max(worst_margin, float(mar[moved].max()))
    rep.update(tokens_moved_gt_1_frame=n_moved, worst_dtw_margin_of_a_moved_token=worst_margin,
               worst_path_excess_frac=worst_excess_frac, surface_rel_l2=surf_rel, surface_maxabs=surf_abs)
    if dump is not None:
        dump["matrix"] = np.stack(mats)
    if ctrl is not None and dtype in ("bf16", "f16") and all(int(c) in streams for c in ctrl["clips"]):
        # a float16 context is held to the reference's own float16 arithmetic, a bf16 one to its bf16 arithmetic
        check_vs_control(z, ctrl, {b: m for b, m in zip(streams, mats)}, ts, rep, problems, tag="fp16" if dtype == "f16" else "bf16")
    if dtype == "f32" and dev.max() > 0.0201:
        problems.append(f"strict f32: token timestamps deviate by {dev.max():.3f} s")
    if surf_rel > bounds["surface_rel"]:
        problems.append(f"alignment surface rel-L2 {surf_rel:.4f} > {bounds['surface_rel']}")
    if worst_excess_frac > bounds["excess_frac"]:
        problems.append(f"engine path costs {worst_excess_frac:.5f} of the optimum more on the reference surface (> {bounds['excess_frac']})")
    if rep["token_ts_within_1_frame_frac"] < bounds["within_1_frame"]:
        problems.append(f"only {rep['token_ts_within_1_frame_frac']:.3f} of the token timestamps within one frame (< {bounds['within_1_frame']})")
    return rep, problems


def run_case(name, dtype, logit_tol, enc_tol, top_abs, ts_bounds, check_ids=True, margin_mult=4.0, min_argmax_agreement=None):
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
        top1_bad = []   # (stream, step, golden margin) of a wrong arg-max above the margin bound
        bound_steps = [0, 0]   # steps on which the top-1 rule binds / all steps

        slice_worst = [0.0]

        fp8g = load_fp8(name) if dtype == "fp8a16" else None
        fp8_stat = {"rel": 0.0, "top": 0.0, "flips": []}

        def teacher(ids, tops, top_idx, sample, sl_proj=None, sl_norm=None, ext=None):
            eng.decoder_reset(B)
            worst_rel, worst_top, flips = 0.0, 0.0, []
            for s in range(ids.shape[1]):
                lg = eng.decode_step(ids[:, s].tolist()).cpu().numpy()
                worst_rel = max(worst_rel, rel_l2(lg[:, ::stride], sample[:, s]))
                if ext is not None:     # the same step against the external W8A16 golden, on its clips
                    cc = [int(c) for c in ext["clips"]]
                    fp8_stat["rel"] = max(fp8_stat["rel"], rel_l2(lg[cc][:, ::stride], ext["logits_sample"][:, s]))
                    for i, c in enumerate(cc):
                        fp8_stat["top"] = max(fp8_stat["top"], float(np.abs(lg[c, top_idx[c, s]] - ext["logits_top"][i, s]).max()))
                        if int(lg[c].argmax()) != int(ext["argmax"][i, s]):
                            fp8_stat["flips"].append((c, s))
                if sl_proj is not None:
                    # WHOLE-ROW evidence per sampler slice (oracle/make_golden_full.py: slice_evidence): the random-sign projection
                    # of every one of the 32 vocabulary slices moves by at most (relative rounding error) x (slice norm); a bug
                    # confined to a slice - which the strided sample of the row can miss - moves it by the norm of what it broke
                    from oracle.make_golden_full import slice_evidence

                    pe, _ = slice_evidence(lg)
                    dev = np.abs(pe - sl_proj[:, s]) / np.maximum(sl_norm[:, s], 1e-6)
                    slice_worst[0] = max(slice_worst[0], float(dev.max()))
                for b in range(B):
                    worst_top = max(worst_top, float(np.abs(lg[b, top_idx[b, s]] - tops[b, s]).max()))
                    margin = tops[b, s, 0] - tops[b, s, 1]
                    bound_steps[1] += 1
                    bound_steps[0] += int(margin > margin_mult * top_abs)
                    if margin > margin_mult * top_abs:
                        if int(lg[b].argmax()) != int(top_idx[b, s, 0]):
                            top1_bad.append((b, s, float(margin)))
                    elif int(lg[b].argmax()) != int(top_idx[b, s, 0]):
                        flips.append((b, s, float(margin)))
            return worst_rel, worst_top, flips

        seq = z["sequences"].astype(np.int64)
        has_slices = "logits_slice_proj" in z.files
        rep["greedy_path_logits_rel_l2"], rep["greedy_path_top8_maxabs"], rep["greedy_path_subm_flips"] = teacher(
            seq[:, :-1], z["logits_top"], z["logits_top_idx"], z["logits_sample"],
            z["logits_slice_proj"] if has_slices else None, z["logits_slice_norm"] if has_slices else None, ext=fp8g)
        # The same statistic for the reference's OWN reduced-precision arithmetic (round 6; oracle/make_golden_ctrl.py stores HF-bf16's and
        # HF-fp16's teacher-forced arg-max along the same fp32 path): an engine flip is legitimate where the reference, cast to the
        # same type, flips as often - asserted as a count (1.25 x the control's + a Poisson allowance, below)
        # and reported per clip, with the number of control clips each of the two decodes identically to fp32.
        ctrl_am = load_ctrl(name)
        tag = {"bf16": "bf16", "f16": "fp16"}.get(dtype)
        if ctrl_am is not None and tag is not None and f"{tag}_argmax" in ctrl_am.files:
            cc = [int(c) for c in ctrl_am["clips"]]
            am = ctrl_am[f"{tag}_argmax"]
            top1 = z["logits_top_idx"][cc][:, : am.shape[1], 0]
            hf_fl = [(c, int(s_)) for i, c in enumerate(cc) for s_ in np.nonzero(am[i] != top1[i])[0]]
            en_fl = [(b, s_) for (b, s_, _m) in rep["greedy_path_subm_flips"] if b in cc and s_ < am.shape[1]]
            rep[f"argmax_flips_on_ctrl_clips(engine, HF-{tag})"] = (len(en_fl), len(hf_fl))
            rep[f"ctrl_clips_argmax_identical_to_fp32(engine, HF-{tag}, of)"] = (
                sum(1 for c in cc if not any(b == c for b, _ in en_fl)), sum(1 for c in cc if not any(b == c for b, _ in hf_fl)), len(cc))
            mg = z["logits_top"][:, :, 0] - z["logits_top"][:, :, 1]
            rep[f"max_margin_of_a_flip(engine, HF-{tag})"] = (max([float(mg[b, s_]) for b, s_ in en_fl], default=0.0),
                                                              max([float(mg[c, s_]) for c, s_ in hf_fl], default=0.0))
            rep[f"argmax_flips_per_clip(engine over all {B} clips, HF-{tag} over its {len(cc)})"] = (
                round(len(rep["greedy_path_subm_flips"]) / B, 3), round(len(hf_fl) / len(cc), 3))
            flips_vs_ctrl = (len(en_fl), len(hf_fl))
        else:
            flips_vs_ctrl = None
        # A11 given IDENTICAL ids by construction: the teacher-forced pass left the alignment heads' softmax rows of the
        # reference's own greedy path in the context, for every stream (whether or not the free-running loop below stays on it)
        Lg = seq.shape[1]
        ts_tf = eng.token_timestamps(B, 3, Lg, [2 * T] * B)
        if DUMP:
            dump.update(ts_teacher_forced=ts_tf)
        ts_rep, problems = check_timestamps(z, eng, ts_tf, list(range(B)), dtype, Lg - 1, ts_bounds, dump if DUMP else None,
                                            ctrl=load_ctrl(name))
        rep.update(ts_rep)
        if fp8g is not None:
            # W8A16 against ITS OWN model definition (round 6): what is left between the engine and "HF float32 over the de-quantised
            # parameters" is the engine's bf16 activation arithmetic (+ the composed cross-query weight), so the yardstick is what HF-bf16
            # itself loses against HF-fp32 on the same clips (ctrl_<case>.npz): logits rel-L2 / top-8 / alignment surface within
            # 1.25 x, teacher-forced arg-max flips within the Poisson allowance used for bf16 above
            ctrl8 = load_ctrl(name)
            cc = [int(c) for c in fp8g["clips"]]
            assert ctrl8 is not None and [int(c) for c in ctrl8["clips"]] == cc
            Lm1 = fp8g["argmax"].shape[1]
            gs, gt = z["logits_sample"][cc][:, :Lm1], z["logits_top"][cc][:, :Lm1]
            hf_rel = max(rel_l2(ctrl8["bf16_logits_sample"][:, s_], gs[:, s_]) for s_ in range(Lm1))
            hf_top = float(np.abs(ctrl8["bf16_logits_top"] - gt).max())
            hf_flips = int((ctrl8["bf16_argmax"] != z["logits_top_idx"][cc][:, :Lm1, 0]).sum())
            Mg = z["dtw_matrix"]
            from tests.util import alignment_matrix as _am
            al8 = eng.get_alignment(B, Lg - 1)
            e_surf = [rel_l2(_am(al8[c], 3), fp8g["dtw_matrix"][i]) for i, c in enumerate(cc)]
            h_surf = [rel_l2(ctrl8["bf16_dtw_matrix"][i], Mg[c]) for i, c in enumerate(cc)]
            e_out = [int((np.abs(ts_tf[c] - fp8g["token_timestamps"][i]) > 0.0201).sum()) for i, c in enumerate(cc)]
            rep["vs_w8a16_model(engine fp8a16 vs HF-fp32 over the W8A16 parameters | HF-bf16 vs HF-fp32)"] = dict(
                logits_rel_l2=(round(fp8_stat["rel"], 5), round(hf_rel, 5)), top8_maxabs=(round(fp8_stat["top"], 4), round(hf_top, 4)),
                argmax_flips=(len(fp8_stat["flips"]), hf_flips), surface_rel_l2=([round(x, 4) for x in e_surf], [round(x, 4) for x in h_surf]),
                tokens_outside_1_frame_vs_w8a16_model=e_out, tokens_per_clip=int(ts_tf.shape[1]))
            if fp8_stat["rel"] > 1.25 * hf_rel:
                problems.append(f"fp8a16 logits rel-L2 {fp8_stat['rel']:.4f} vs the W8A16 model > 1.25 x HF-bf16's own {hf_rel:.4f}")
            if fp8_stat["top"] > 1.25 * hf_top:
                problems.append(f"fp8a16 top-8 values {fp8_stat['top']:.4f} off the W8A16 model > 1.25 x HF-bf16's own {hf_top:.4f}")
            if len(fp8_stat["flips"]) > 1.25 * hf_flips + 2.0 * max(1.0, hf_flips) ** 0.5:
                problems.append(f"fp8a16: {len(fp8_stat['flips'])} arg-max flips against the W8A16 model; HF-bf16 against HF-fp32: {hf_flips}")
            for c, e_, h_ in zip(cc, e_surf, h_surf):
                if e_ > 1.25 * h_:
                    problems.append(f"clip {c}: fp8a16 alignment surface {e_:.4f} off the W8A16 model's > 1.25 x HF-bf16's own {h_:.4f}")
        rep["rand_path_logits_rel_l2"], rep["rand_path_top8_maxabs"], rep["rand_path_subm_flips"] = teacher(
            z["rand_ids"].astype(np.int64), z["rand_logits_top"], z["rand_logits_top_idx"], z["rand_logits_sample"],
            z["rand_logits_slice_proj"] if has_slices else None, z["rand_logits_slice_norm"] if has_slices else None)
        # (flips are rare, independent events - 0.7-0.9 per 162-step clip for HF-bf16 and for the engine alike: the bound is 1.25 x the
        #  control's count plus two standard deviations of a Poisson count of that size; on the 16-clip case round 6 measures 6 vs 3 on
        #  the four control clips and 14 on all 16 clips = 0.875 per clip against HF-bf16's 0.75)
        if flips_vs_ctrl is not None and flips_vs_ctrl[0] > 1.25 * flips_vs_ctrl[1] + 2.0 * max(1.0, flips_vs_ctrl[1]) ** 0.5:
            problems.append(f"{flips_vs_ctrl[0]} teacher-forced arg-max flips on the control clips; the reference's own {tag} arithmetic has {flips_vs_ctrl[1]}")
        if has_slices:
            rep["slice_projection_worst_dev_over_slice_norm"] = slice_worst[0]
            if slice_worst[0] > 4 * logit_tol:     # rounding noise adds up like a random walk under the random signs: ~ the rel-L2 itself
                problems.append(f"a vocabulary slice's random-sign projection is off by {slice_worst[0]:.4f} of the slice norm (> {4 * logit_tol})")
        rep["top1_rule_binds_on_frac_of_steps"] = round(bound_steps[0] / max(1, bound_steps[1]), 3)
        # id-level statement that CAN fail whatever the margins are (round-5 advice, for the flavour whose top-1 rule is switched off):
        # the share of teacher-forced steps (both passes) whose arg-max is the fp32 reference's
        n_flip = len(rep["greedy_path_subm_flips"]) + len(rep["rand_path_subm_flips"]) + len(top1_bad)
        rep["argmax_agreement_frac"] = round(1.0 - n_flip / max(1, bound_steps[1]), 4)
        if min_argmax_agreement is not None and rep["argmax_agreement_frac"] < min_argmax_agreement:
            problems.append(f"arg-max identical to fp32 on only {rep['argmax_agreement_frac']:.3f} of the teacher-forced steps (< {min_argmax_agreement})")
        if top1_bad:
            problems.append(f"top-1 differs above the margin bound at (stream, step, margin) {top1_bad[:8]}")
        if not (rep["greedy_path_logits_rel_l2"] < logit_tol and rep["rand_path_logits_rel_l2"] < logit_tol):
            problems.append(f"logits rel-L2 above {logit_tol}")
        if not (rep["greedy_path_top8_maxabs"] < top_abs and rep["rand_path_top8_maxabs"] < top_abs):
            problems.append(f"top-8 logit values off by more than {top_abs}")

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
                if m > margin_mult * top_abs:
                    problems.append(f"stream {b} diverges at position {p} where the golden margin is {m}")
            rep["first_divergence(pos, golden_margin)"] = first_div
            rep["min_golden_margin"] = float(margins[:, 2 : seq.shape[1] - 1].min())
            same = [b for b in range(B) if first_div[b] is None and got.shape[1] == seq.shape[1]]
            rep["streams_with_identical_ids"] = len(same)
            ts = eng.token_timestamps(B, 3, got.shape[1], [2 * T] * B)
            if DUMP:
                dump.update(ids=got, ts=ts)
            if same:
                # same ids, same arithmetic (captured step graph or not): the free-running loop's timestamps ARE the teacher-forced ones
                if not np.array_equal(ts[same], ts_tf[same]):
                    problems.append("free-running and teacher-forced token timestamps differ on identical ids")
    finally:
        eng.close()
        if DUMP and dump:
            os.makedirs(DUMP, exist_ok=True)
            np.savez_compressed(os.path.join(DUMP, f"{name}_{dtype}.npz"), **dump)
    from thewhisper_amd.build import source_digest

    rep["kernel_source_sha256"] = source_digest()      # which kernels these figures belong to (bench.py: full_depth_parity)
    print(f"\nFULLDEPTH {name} {dtype}: " + ", ".join(f"{k}={v}" for k, v in rep.items()))
    assert not problems, (name, dtype, problems, rep)
    return rep


# bounds: (logits rel-L2, encoder rel-L2, top-8 abs)
# ts_bounds: alignment-surface rel-L2, engine-path excess cost on the reference surface (fraction of the optimum; a quarter of the
# surface tolerance in the reduced-precision contexts: how far from optimal a path can be is a matter of how far the surface may be
# off - the first 35-token clip of the 2-clip case sits at 0.018 with a surface error of 0.105, BELOW HF-bf16's own 0.114), floor
# of the fraction of token timestamps within one frame (a regression alarm, not the parity statement - see check_timestamps)
# Measured on the MI355X (profiles/r03_gpu_tests_full_depth.log), bounds = ~1.5 x the worst case:
#   bf16: logits rel-L2 0.0095-0.0161, top-8 0.035-0.073, alignment surface rel-L2 0.032 (turbo) - 0.123 (32 decoder layers,
#         16 clips), engine-path excess 1e-4 - 7.2e-3 of the optimum, 77 % (16 clips x 159 tokens) - 94 % within one frame
#   fp8 : logits 0.084-0.087, top-8 0.28-0.29, surface 0.34-0.36 (stable); the PATH over that surface is not: two builds whose bf16
#         encoders differ in the last bit gave excess 0.050 / 0.123 and 57 % / 34 % of the tokens within one frame (worst 1.2 / 7.2 s).
#         The surface error comes from the MXFP8 weight / activation quantisation of the network state, not from the e4m3 cross
#         keys (oracle experiment), so only the surface is bounded tightly; the path figures are alarms.
F32 = dict(logit_tol=2e-4, enc_tol=2e-4, top_abs=2e-3, ts_bounds=dict(surface_rel=1e-4, excess_frac=1e-6, within_1_frame=1.0))
# (within_1_frame is a gross alarm only since round 4: what the bf16 engine may lose is bounded RELATIVE to the reference's own bf16
#  arithmetic by check_vs_control - HF-bf16 itself is at 0.53-0.94 within one frame on these cases)
BF16 = dict(logit_tol=3e-2, enc_tol=3e-2, top_abs=0.12, ts_bounds=dict(surface_rel=0.18, excess_frac=0.045, within_1_frame=0.45))
# float16 contexts (round 4): HF-fp16 itself is at encoder 1.6e-3, logits 2.1e-3, top-8 0.009, surface 0.005-0.018 against fp32
F16 = dict(logit_tol=5e-3, enc_tol=4e-3, top_abs=0.02, ts_bounds=dict(surface_rel=0.03, excess_frac=0.0075, within_1_frame=0.6))
# MXFP8 decoder weights + e4m3 cross-K/V (BASELINE config 5): the encoder is bf16, so its bound is bf16's.
# Top-1 rule: every logit within `top_abs` of the reference means the arg-max can only change where the reference margin is below
# 2 x top_abs, so the rule is "identical wherever the golden margin exceeds 2 x top_abs" (the 4 x of the other dtypes made it bind on
# 0-3 % of the steps of the 15 s goldens: vacuous); the fraction of steps it binds on is printed and asserted to be >= a third.
#   fp8a8  (W8A8, scaled fp8 MFMA): logits 0.084-0.087, top-8 0.28-0.29 measured in round 3
#   fp8a16 (W8A16, weights widened to bf16, activations not quantised): bounds below are round 4's first measurement x 1.5
# Measured in round 4 (profiles/r04_gpu_tests_full_depth.log), 15 s goldens (1 clip x 32 tokens / 4 clips x 128 tokens); bounds ~1.5 x:
#            logits rel-L2   top-8         alignment surface rel-L2   path excess     within one frame
#   bf16     0.016           0.068-0.076   0.104-0.115                0.003-0.005     0.80 / 0.53   (HF-bf16 itself: 0.94 / 0.53)
#   fp8a16   0.063-0.066     0.20-0.21     0.145-0.163                0.009-0.014     0.66 / 0.43
#   fp8a8    0.082-0.085     0.28-0.34     0.37-0.41                  0.024-0.127     0.57 / 0.39
# Round 5: bounds = 1.25 x the worst value measured in round 4 (the previous 1.5-2 x left room for a regression of half the effect):
#   fp8a16 worst: logits 0.0662, top-8 0.2142, surface 0.1585, excess 0.0137, within one frame 0.50
#   fp8a8  worst: logits 0.0861, top-8 0.3950 (random-token pass), surface 0.3991, excess 0.1566, within one frame 0.385
# W8A8's top-1 rule ("arg-max identical wherever the golden margin exceeds 2 x top_abs" = 1.0) binds on 0-0.5 % of the steps of these
# goldens, i.e. it asserts nothing (round-3 and round-4 reviews): it is SWITCHED OFF for that flavour (margin_mult = inf: every
# arg-max change is reported as a sub-margin flip, none is an error) rather than kept as a rule that cannot fail; what W8A8 is held
# to are the logits / top-8 / surface / excess bounds.  W8A16 - the flavour that ships as dtype="fp8" - keeps the rule, which binds on
# a third of all steps at top_abs = 0.27 (asserted >= 0.30 below).
# Round 6: W8A8's within-one-frame fraction is NOT a stable statistic.  The round's change of the long-K projection's summation order
# (sixteen K slices added in pairs, k_decode.hip - float32 ulps) moved it from 0.385 to 0.227 on the 4-clip golden while logits rel-L2
# (0.0806), surface rel-L2 (0.399 -> 0.396) and path excess (0.157 -> 0.166) stayed where they were: with 40 % error on the alignment
# surface the DTW path is decided by noise.  The floor is kept only as a sanity bound (0.15); the bounds that hold the flavour are the
# logits / top-8 / surface / excess ones.  (W8A16, the flavour that ships, moved 0.50 -> see profiles/r06_gpu_tests_full_depth.log.)
# (round 6) ... and an id-level floor that does not depend on margins: arg-max identical to fp32 on >= 75 % of all teacher-forced steps
# (measured: 0.82 on the 1-clip golden - 12 flips on 68 steps -, 0.91 on the 4-clip one; an arg-max regression of the
# W8A8 path now fails the suite)
FP8A8 = dict(logit_tol=0.108, enc_tol=3e-2, top_abs=0.5, margin_mult=float("inf"), min_argmax_agreement=0.75, ts_bounds=dict(surface_rel=0.5, excess_frac=0.2, within_1_frame=0.15))
FP8A16 = dict(logit_tol=0.083, enc_tol=3e-2, top_abs=0.27, margin_mult=2.0, ts_bounds=dict(surface_rel=0.2, excess_frac=0.0175, within_1_frame=0.4))


# ordered so that consecutive cases share the (6 GB, ~20 s to generate) seeded state dict
CASES = [("full_turbo_c30", "bf16"), ("full_turbo_c30", "f16"), ("full_turbo_c30", "f32"),
         ("full_large-v3_c10", "bf16"), ("full_large-v3_c10", "f16"), ("full_large-v3_c10", "f32"),
         ("full_large-v3_c10_b16", "bf16"), ("full_large-v3_c10_b16", "f16"), ("full_large-v3_c10_b16", "f32"),
         ("full_large-v3_c15", "bf16"), ("full_large-v3_c15", "f16"),
         ("full_large-v3_c15", "fp8a8"), ("full_large-v3_c15", "fp8a16"),
         # 20 s chunks (T = 1000: 15 full 64-key tiles + 40, the two-chunk cross attention): the fourth length the reference advertises
         ("full_large-v3_c20", "bf16"), ("full_large-v3_c20", "f16"), ("full_large-v3_c20", "f32"),
         # config 5 at the length its driver-timed leg decodes: 4 clips x 128 new tokens at 15 s
         ("full_large-v3_c15_b4", "bf16"), ("full_large-v3_c15_b4", "fp8a8"), ("full_large-v3_c15_b4", "fp8a16")]


@pytest.mark.parametrize("name,dtype", CASES)
def test_full_depth(name, dtype):
    if not os.path.exists(os.path.join(GOLD, f"{name}.npz")):
        pytest.skip(f"{name}.npz not generated (oracle/make_golden_full.py)")
    rep = run_case(name, dtype, **{"f32": F32, "bf16": BF16, "f16": F16, "fp8a8": FP8A8, "fp8a16": FP8A16}[dtype])
    if dtype.startswith("fp8"):
        # the fp8 top-1 rule is not vacuous for the flavour that ships (`dtype="fp8"` = W8A16): it binds on ~a third of ALL
        # teacher-forced steps (0.32-0.38 measured; the random-token pass, whose margins are ~0.1, included).  W8A8's larger logit
        # error (top-8 0.28-0.34 -> rule at 1.0) leaves it vacuous there, as the round-3 review found: reported, not asserted.
        if dtype == "fp8a16":
            assert rep["top1_rule_binds_on_frac_of_steps"] >= 0.30, rep
    if dtype == "f16":
        # the id-identity claim on the 16-bit dtype that can carry it (the reference's streaming default): on EVERY clip of EVERY full-depth
        # case the free-running greedy ids equal the HF fp32 reference's to the end - 16 of 16 on the headline-shaped case
        # (bench.py reads this figure from the committed log: parity_full_depth.f16.ids_identical_clips)
        assert all(d is None for d in rep["first_divergence(pos, golden_margin)"]), rep
        assert rep["streams_with_identical_ids"] == len(rep["first_divergence(pos, golden_margin)"]), rep
    if dtype == "f32" and rep["min_golden_margin"] > 4 * F32["top_abs"]:
        # strict mode: every decision margin on these clips is above the bound, so the ids must be identical outright
        assert all(d is None for d in rep["first_divergence(pos, golden_margin)"]), rep


@pytest.mark.parametrize("dtype,n_streams", [("f32", 64), ("f16", 64), ("f32", 32)])
def test_full_depth_64_streams_follow_the_16_clip_golden(dtype, n_streams):
    """The kernels for MORE than 16 streams (operand-ring projections: four groups of 16 streams per weight pass, k_decode.hip) at full
    depth against the reference: the 16 clips of the headline-shaped golden (HF fp32, 32 + 32 layers, 160 free-running greedy tokens
    with the timestamp grammar) decoded FOUR TIMES side by side in one 64-stream context.  Strict f32: every one of the 64 rows
    reproduces its clip's reference ids to the end and the teacher-forced logits to 2e-4; float16: ids identical too (as the 16-stream
    context does, test_full_depth), logits to 5e-3.  Rows of different groups that carry the same clip are bit-identical."""
    name = "full_large-v3_c10_b16"
    if not os.path.exists(os.path.join(GOLD, f"{name}.npz")):
        pytest.skip(f"{name}.npz not generated (oracle/make_golden_full.py)")
    z, dims, w, pcm, heads = load_case(name)
    T, B0 = 50 * int(z["chunk_s"]), pcm.shape[0]
    R = n_streams // B0      # 64 streams: four groups of 16 per weight pass; 32: two (the other instantiation of the ring kernels)
    B = R * B0
    stride = int(z["logit_stride"]) if "logit_stride" in z.files else STRIDE
    eng = make_engine(dims, w, T=T, max_batch=B, dtype=dtype, heads=heads, use_graph=True)
    try:
        mel = eng.logmel(torch.from_numpy(np.tile(pcm, (R, 1))).cuda(), out_dtype=torch.float32)
        eng.encode(mel)
        eng.cross_kv(B)
        seq = np.tile(z["sequences"].astype(np.int64), (R, 1))
        # teacher-forced along the reference's greedy path: a strided sample of every logits row against the golden sample
        eng.decoder_reset(B)
        worst = 0.0
        for s in range(min(24, seq.shape[1] - 1)):
            lg = eng.decode_step(seq[:, s].tolist()).cpu().numpy()
            ref = np.tile(z["logits_sample"][:, s], (R, 1))
            worst = max(worst, rel_l2(lg[:, ::stride], ref))
            for r in range(1, R):    # the same clip in another group of 16 streams
                assert np.array_equal(lg[:B0], lg[r * B0 : (r + 1) * B0]), (s, r)
        assert worst < (2e-4 if dtype == "f32" else 5e-3), worst
        # free-running greedy ids: all 64 rows on their clip's reference path
        prompt = np.tile(np.array(PROMPT, dtype=np.int32), (B, 1))
        got = eng.generate_greedy(prompt, max_new_tokens=int(z["max_new"]), timestamps=True, want_alignment=True)["sequences"]
        L = min(got.shape[1], seq.shape[1])
        same = [b for b in range(B) if got.shape[1] == seq.shape[1] and np.array_equal(got[b, :L], seq[b, :L])]
        print(f"\nFULLDEPTH64 {name} {dtype}: {B} streams = 16 golden clips x {R}, teacher-forced logits rel-L2 {worst:.3e}, "
              f"streams_with_identical_ids={len(same)} of {B}")
        assert len(same) == B, sorted(set(range(B)) - set(same))
    finally:
        eng.close()
