"""The reduced-precision CONTROL vectors (oracle/make_golden_ctrl.py -> tests/golden/ctrl_full_*.npz): what the reference's own
arithmetic (HF transformers on CPU, R:thestage_speechkit/nvidia/asr_pipeline.py:57-60) loses against its fp32 self when the
model is cast to bf16 (the engine's production dtype) or fp16 (the reference's streaming default,
R:thestage_speechkit/streaming/streaming_pipeline.py:369-370), teacher-forced along the fp32 greedy path.  These numbers are
the yardstick tests/test_gpu_full_depth.py holds the bf16 engine to (<= 1.25 x HF-bf16's own surface error and moved tokens);
here they are checked for consistency with the fp32 goldens and their magnitudes are pinned, so a regenerated control that
drifted (other transformers / torch build) is noticed on CPU."""
import os

import numpy as np
import pytest

from tests.util import rel_l2

GOLD = os.path.join(os.path.dirname(__file__), "golden")
#        case                      HF-bf16 surface rel-L2 (min, max)   HF-fp16 (max)
CASES = {"full_large-v3_c10": ((0.08, 0.15), 0.03), "full_large-v3_c10_b16": ((0.09, 0.16), 0.03),
         "full_large-v3_c15": ((0.08, 0.15), 0.03), "full_turbo_c30": ((0.02, 0.06), 0.01), "full_large-v3_c20": ((0.08, 0.16), 0.03)}


@pytest.mark.parametrize("name", list(CASES))
def test_control_is_consistent_with_the_fp32_golden(name):
    z = np.load(os.path.join(GOLD, f"{name}.npz"))
    c = np.load(os.path.join(GOLD, f"ctrl_{name}.npz"))
    clips = [int(x) for x in c["clips"]]
    L = z["sequences"].shape[1]
    (lo, hi), f16_hi = CASES[name]
    for tag in ("bf16", "fp16"):
        m, ts = c[f"{tag}_dtw_matrix"], c[f"{tag}_token_timestamps"]
        assert m.shape == z["dtw_matrix"][clips].shape and ts.shape == (len(clips), L)
        assert (ts[:, :3] == 0).all() and np.isfinite(m).all()
        # same teacher-forced path, same top-8 index set: logits close to the fp32 golden's
        top = np.abs(c[f"{tag}_logits_top"] - z["logits_top"][clips][:, : L - 1]).max()
        assert top < (0.12 if tag == "bf16" else 0.02), (tag, top)
        assert rel_l2(c[f"{tag}_enc_rows"], z["enc_rows"][clips]) < (3e-2 if tag == "bf16" else 4e-3)
    for i, b in enumerate(clips):
        rb = rel_l2(c["bf16_dtw_matrix"][i], z["dtw_matrix"][b])
        rf = rel_l2(c["fp16_dtw_matrix"][i], z["dtw_matrix"][b])
        assert lo < rb < hi, (name, b, rb)        # the reference's OWN bf16 surface error: ~0.11-0.13 at 32 decoder layers, 0.037 turbo
        assert rf < f16_hi and rf < rb / 4, (name, b, rf)   # fp16 (10 mantissa bits) is ~7x closer than bf16 (7 bits)


def test_hf_bf16_itself_moves_token_timestamps():
    """"identical +-0.02 s" does not hold for the reference's own arithmetic in bf16 either: on the 16-clip case (4 control clips x
    163 tokens) HF-bf16 moves a quarter of the tokens by more than one frame, the worst by over a second; HF-fp16 none."""
    z = np.load(os.path.join(GOLD, "full_large-v3_c10_b16.npz"))
    c = np.load(os.path.join(GOLD, "ctrl_full_large-v3_c10_b16.npz"))
    g = z["token_timestamps"][[int(x) for x in c["clips"]]]
    dev_b, dev_f = np.abs(c["bf16_token_timestamps"] - g), np.abs(c["fp16_token_timestamps"] - g)
    assert 0.15 < (dev_b > 0.0201).mean() < 0.45 and dev_b.max() > 0.5
    assert (dev_f > 0.0201).mean() < 0.02


def test_hf_bf16_itself_flips_greedy_decisions():
    """Round 6: the control files carry HF-bf16's and HF-fp16's OWN teacher-forced arg-max.  "Greedy ids identical" does not hold for
    the reference's arithmetic cast to bf16 either: it flips 1-3 decisions per 130-162-token clip (every flip at a fp32 top-1/top-2
    margin below 0.05) and decodes at most half of the control clips identically to fp32, while HF-fp16 flips none.  This is the
    yardstick `tests/test_gpu_full_depth.py` holds the engine's sub-margin flips to."""
    n_flip = {"bf16": 0, "fp16": 0}
    ident = {"bf16": 0, "fp16": 0}
    n_clips = 0
    for name in list(CASES) + ["full_large-v3_c15_b4"]:
        z = np.load(os.path.join(GOLD, f"{name}.npz"))
        c = np.load(os.path.join(GOLD, f"ctrl_{name}.npz"))
        clips = [int(x) for x in c["clips"]]
        n_clips += len(clips)
        for tag in ("bf16", "fp16"):
            am = c[f"{tag}_argmax"]
            assert am.shape == (len(clips), z["sequences"].shape[1] - 1)
            top = z["logits_top_idx"][clips][:, : am.shape[1], 0]
            lt = z["logits_top"][clips][:, : am.shape[1]]
            fl = am != top
            n_flip[tag] += int(fl.sum())
            ident[tag] += int((~fl.any(axis=1)).sum())
            if fl.any():       # a flip only where fp32 itself barely decides
                assert float((lt[..., 0] - lt[..., 1])[fl].max()) < 0.06, (name, tag)
    print(f"\nCONTROL arg-max: HF-bf16 flips {n_flip['bf16']} decisions on {n_clips} clips ({ident['bf16']} clips identical to fp32), "
          f"HF-fp16 {n_flip['fp16']} ({ident['fp16']} identical)")
    assert n_flip["fp16"] == 0 and ident["fp16"] == n_clips
    assert n_flip["bf16"] >= 8 and ident["bf16"] <= n_clips // 2
