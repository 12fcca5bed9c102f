"""Reduced-precision CONTROL vectors: what the reference's own arithmetic does to the word-timestamp surface in bf16 / fp16.

    python -m oracle.make_golden_ctrl [case ...]        (CPU, this container; minutes per case; never on the GPU box)

TEST INFRASTRUCTURE ONLY.  tests/golden/full_*.npz hold HF *float32* results.  The engine ships bf16, and the reference's
streaming default is float16 (R:thestage_speechkit/streaming/streaming_pipeline.py:369-370), so "how far may a bf16 engine be
from the fp32 surface" needs a yardstick that is not the engine's own: the SAME HF model (R:thestage_speechkit/nvidia/
asr_pipeline.py:57-60) cast with ``.to(torch.bfloat16)`` / ``.to(torch.float16)`` on CPU, teacher-forced along the fp32
greedy path of the golden file (no cache: one pass, as make_golden_full.py does), its cross-attention rows handed to HF's
own ``_extract_token_timestamps`` (z-score, median filter and head mean therefore run in the reduced dtype, exactly as they
would inside ``generate``), with ``_dynamic_time_warping`` spied on for the surface.

Stored per case under tests/golden/ctrl_<case>.npz, per dtype tag (bf16, fp16), for the first clips of the case:
  <tag>_dtw_matrix        [clips, new tokens - 1, T]   the surface HF-<tag> hands (negated) to its DTW
  <tag>_token_timestamps  [clips, L]
  <tag>_logits_top        [clips, L - 1, 8]            HF-<tag> logits at the fp32 golden's top-8 indices
  <tag>_logits_sample     [clips, L - 1, V / stride]   the same strided sample of every logits row the fp32 golden keeps
  <tag>_enc_rows          encoder-state samples (same strides as the fp32 golden)
  <tag>_argmax            [clips, L - 1]               HF-<tag>'s OWN teacher-forced arg-max over the whole vocabulary (round 6): where it
                                                        differs from the fp32 golden's top-1 the reference's own reduced-precision
                                                        arithmetic flips a decision - the yardstick for the engine's sub-margin flips
tests/test_golden_ctrl.py summarises them on CPU; tests/test_gpu_full_depth.py holds the engine to 1.25 x these errors.
"""
from __future__ import annotations

import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tests", "golden")

from . import hf_reference as hr  # noqa: E402
from . import whisper_oracle as wo  # noqa: E402
from .make_golden_full import ENC_DSTRIDE, ENC_TSTRIDE, PROMPT  # noqa: E402

# case -> clip indices of the fp32 golden that get a control (the 16-clip case: one of each audio kind)
CTRL = {
    "full_large-v3_c10": [0, 1],
    "full_large-v3_c10_b16": [0, 1, 2, 3],
    "full_large-v3_c15": [0],
    "full_large-v3_c20": [0, 1],
    "full_large-v3_c15_b4": [0, 1, 2, 3],
    "full_turbo_c30": [0],
}
DTYPES = {"bf16": torch.bfloat16, "fp16": torch.float16}


class _Outputs(dict):
    """What ``_extract_token_timestamps`` reads of a generate output: ``.sequences``, ``.cross_attentions``, ``"beam_indices" in``."""

    __getattr__ = dict.__getitem__


def run_case(name: str):
    import transformers.models.whisper.generation_whisper as gw

    z = np.load(os.path.join(OUT, f"{name}.npz"))
    clips = CTRL[name]
    preset, chunk_s = str(z["preset"]), int(z["chunk_s"])
    dims = wo.PRESETS[preset]
    T = 50 * chunk_s
    stride = int(z["logit_stride"])
    seq = torch.from_numpy(z["sequences"][clips].astype(np.int64))
    top_idx = torch.from_numpy(z["logits_top_idx"][clips].astype(np.int64))
    nB, L = seq.shape
    t0 = time.time()
    w = wo.make_weights(dims, int(z["weight_seed"]), scale=float(z["weight_scale"]), q_gain=float(z["q_gain"]))
    model = hr.build_hf_model(dims, w)
    del w
    hr.patch_chunk_length(model, chunk_s)
    model.config._attn_implementation = "eager"   # output_attentions (HF forces the same for word timestamps, generation_whisper.py:706-707)
    fe = hr.build_feature_extractor(dims, chunk_s)
    pcm = np.stack([wo.synth_audio(16000 * chunk_s, int(z["clip_seeds"][i]), str(z["clip_kinds"][i])) for i in clips])
    mel = fe([p for p in pcm], sampling_rate=16000, return_tensors="pt").input_features
    heads = [tuple(int(x) for x in h) for h in z["alignment_heads"]]
    print(f"[{name}] model ready in {time.time() - t0:.0f} s; clips {clips}, L = {L}", flush=True)
    out = dict(clips=np.array(clips), versions=np.array([f"transformers {__import__('transformers').__version__}", f"torch {torch.__version__}"]))
    for tag, dt in DTYPES.items():
        t0 = time.time()
        model = model.to(dt)   # fp32 -> bf16, then bf16 -> fp16 would round twice: rebuild from fp32 instead (below)
        enc = model.model.encoder(mel.to(dt)).last_hidden_state
        dec = model.model.decoder(input_ids=seq[:, :-1], encoder_hidden_states=enc, output_attentions=True, use_cache=False)
        logits = model.proj_out(dec.last_hidden_state).float()
        cross = tuple(dec.cross_attentions)   # per layer [nB, H, L - 1, T]
        assert cross[0].dtype == dt and tuple(cross[0].shape) == (nB, dims.heads, L - 1, T)
        surfaces = []
        orig = gw._dynamic_time_warping

        def spy(matrix):
            surfaces.append(np.array(matrix, dtype=np.float64))
            return orig(matrix)

        gw._dynamic_time_warping = spy
        try:
            ts = model._extract_token_timestamps(
                _Outputs(sequences=seq, cross_attentions=(cross,)), heads,
                num_frames=torch.tensor([2 * T] * nB), num_input_ids=len(PROMPT))
        finally:
            gw._dynamic_time_warping = orig
        out[f"{tag}_dtw_matrix"] = -np.stack(surfaces).astype(np.float32)
        out[f"{tag}_token_timestamps"] = ts.numpy().astype(np.float32)
        out[f"{tag}_logits_top"] = torch.gather(logits, 2, top_idx[:, : L - 1]).numpy().astype(np.float32)
        out[f"{tag}_logits_sample"] = logits[:, :, ::stride].numpy().astype(np.float32)
        out[f"{tag}_argmax"] = logits.argmax(dim=-1).numpy().astype(np.int32)
        n_flip = int((out[f"{tag}_argmax"] != z["logits_top_idx"][clips][:, : L - 1, 0]).sum())
        out[f"{tag}_enc_rows"] = enc.float()[:, ::ENC_TSTRIDE, ::ENC_DSTRIDE].numpy().astype(np.float32)
        g = z["dtw_matrix"][clips]
        m = out[f"{tag}_dtw_matrix"]
        rel = [float(np.linalg.norm(m[i] - g[i]) / np.linalg.norm(g[i])) for i in range(nB)]
        dev = np.abs(out[f"{tag}_token_timestamps"] - z["token_timestamps"][clips])
        print(f"[{name}] HF-{tag} ({time.time() - t0:.0f} s): surface rel-L2 vs fp32 {np.round(rel, 4).tolist()}, "
              f"within one frame {float((dev <= 0.0201).mean()):.3f}, worst {float(dev.max()):.2f} s, "
              f"top-8 max-abs {float(np.abs(out[f'{tag}_logits_top'] - z['logits_top'][clips][:, : L - 1]).max()):.4f}, "
              f"teacher-forced arg-max differs from the fp32 top-1 on {n_flip} of {nB * (L - 1)} steps", flush=True)
        if tag != list(DTYPES)[-1]:
            # the next dtype starts from the float32 weights again
            w = wo.make_weights(dims, int(z["weight_seed"]), scale=float(z["weight_scale"]), q_gain=float(z["q_gain"]))
            model = hr.build_hf_model(dims, w)
            del w
            hr.patch_chunk_length(model, chunk_s)
            model.config._attn_implementation = "eager"
    np.savez_compressed(os.path.join(OUT, f"ctrl_{name}.npz"), **out)


def main():
    torch.set_grad_enabled(False)
    torch.set_num_threads(os.cpu_count() or 1)
    for name in (sys.argv[1:] or list(CTRL)):
        run_case(name)


if __name__ == "__main__":
    main()
