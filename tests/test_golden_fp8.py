"""Config 5's external pin (oracle/make_golden_fp8.py -> tests/golden/fp8_full_*.npz): the W8A16 context is "the reference's model class
in float32 over the parameters that context stores" - decoder weights de-quantised from MXFP8 with torch's own float8_e4m3fn cast,
LayerNorms folded before quantisation, cross-K/V through an fp8 round trip.  On CPU:

* the stored vectors are consistent with the fp32 goldens and their magnitudes are pinned (a regenerated file that drifted is noticed);
* the repo's numpy restatement of the SAME context (oracle/whisper_oracle.py: OracleWhisperMXFP8(act_quant=False), what the micro-model
  GPU parity cases are held to) agrees with that HF model, run here on the micro preset, to bf16-activation noise - an order of
  magnitude below the effect of the quantisation itself.  The restatement is therefore pinned to the reference's model class too, not
  only to itself.
tests/test_gpu_full_depth.py holds the engine's fp8a16 context to these vectors at full depth (-m gpu)."""
import os

import numpy as np
import pytest
import torch

from oracle import hf_reference as hr
from oracle import whisper_oracle as wo
from tests.util import rel_l2

GOLD = os.path.join(os.path.dirname(__file__), "golden")
torch.set_grad_enabled(False)


@pytest.mark.parametrize("name", ["full_large-v3_c15", "full_large-v3_c15_b4"])
def test_w8a16_golden_is_consistent_with_the_fp32_golden(name):
    z = np.load(os.path.join(GOLD, f"{name}.npz"))
    f = np.load(os.path.join(GOLD, f"fp8_{name}.npz"))
    c = np.load(os.path.join(GOLD, f"ctrl_{name}.npz"))
    clips = [int(x) for x in f["clips"]]
    assert clips == [int(x) for x in c["clips"]]
    L = z["sequences"].shape[1]
    assert f["argmax"].shape == (len(clips), L - 1) and f["token_timestamps"].shape == (len(clips), L)
    assert f["dtw_matrix"].shape == z["dtw_matrix"][clips].shape and np.isfinite(f["dtw_matrix"]).all()
    gs = z["logits_sample"][clips][:, : L - 1]
    q_rel = rel_l2(f["logits_sample"], gs)                       # what the QUANTISATION costs: the W8A16 model vs the unquantised fp32 model
    b_rel = rel_l2(c["bf16_logits_sample"], gs)                  # what bf16 ARITHMETIC costs HF itself
    assert 0.04 < q_rel < 0.08 and 0.008 < b_rel < 0.03 and q_rel > 2.5 * b_rel, (q_rel, b_rel)
    assert 0.02 < float(f["weight_rel_err"]) < 0.035             # e4m3 with a shared block exponent: 2^-4 / sqrt(3) ~ 2.7 %
    top = float(np.abs(f["logits_top"] - z["logits_top"][clips][:, : L - 1]).max())
    assert 0.1 < top < 0.3
    surf = [rel_l2(f["dtw_matrix"][i], z["dtw_matrix"][b]) for i, b in enumerate(clips)]
    assert all(0.08 < s < 0.16 for s in surf), surf
    n_flip = int((f["argmax"] != z["logits_top_idx"][clips][:, : L - 1, 0]).sum())
    print(f"\nW8A16 MODEL {name}: logits rel-L2 vs fp32 {q_rel:.4f} (HF-bf16 arithmetic alone: {b_rel:.4f}), top-8 {top:.3f}, "
          f"arg-max differs on {n_flip} of {len(clips) * (L - 1)} steps, surface rel-L2 {np.round(surf, 3).tolist()}")
    assert n_flip <= 0.05 * len(clips) * (L - 1) + 3


def test_the_numpy_restatement_is_that_hf_model_micro():
    """OracleWhisperMXFP8(act_quant=False) vs HF-float32 over `to_w8a16` parameters on the micro preset, teacher-forced on the same
    ids: logits within bf16-activation noise; both the plain launch sequence and the engine's composed cross query ("cross query
    ahead": W' Wo quantised as ONE matrix) - the composition costs a fraction of what the quantisation does."""
    from oracle.make_golden_fp8 import to_w8a16

    dims = wo.PRESETS["micro"]
    w = wo.make_weights(dims, 0)
    T = 100
    mel = wo.log_mel(np.stack([wo.synth_audio(16000 * 2, s) for s in (0, 1)]), dims.n_mels)
    ids = np.concatenate([np.tile([50258, 50259, 50360], (2, 1)), np.random.default_rng(1).integers(0, 50000, size=(2, 9))], axis=1)

    def hf_logits(transform):
        m = hr.patch_chunk_length(hr.build_hf_model(dims, w), 2)
        m.config._attn_implementation = "eager"
        if transform:
            to_w8a16(m, dims)
        x = torch.from_numpy(mel)
        enc = m.model.encoder(x).last_hidden_state
        dec = m.model.decoder(input_ids=torch.from_numpy(ids), encoder_hidden_states=enc, use_cache=False)
        return m.proj_out(dec.last_hidden_state).numpy()

    exact, model = hf_logits(False), hf_logits(True)
    q_effect = rel_l2(model, exact)
    out = {}
    for ahead in (False, True):
        om = wo.OracleWhisperMXFP8(dims, w, T=T, cross_q_ahead=ahead, act_quant=False)
        cache = om.new_cache(om.encode(mel))
        out[ahead] = rel_l2(om.decode(ids, cache)[0], model)
    print(f"\nW8A16 micro: quantisation effect {q_effect:.4f}; restatement vs the HF model: plain sequence {out[False]:.4f}, composed cross query {out[True]:.4f}")
    assert 0.01 < q_effect < 0.15
    assert out[False] < 0.25 * q_effect and out[True] < 0.5 * q_effect, (out, q_effect)
