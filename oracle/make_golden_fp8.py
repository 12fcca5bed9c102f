"""BASELINE config 5's EXTERNAL pin: the W8A16 context ("fp8") is, by construction, the reference's arithmetic over de-quantised weights.

    python -m oracle.make_golden_fp8 [case ...]        (CPU, this container; minutes per case; never on the GPU box)

TEST INFRASTRUCTURE ONLY.  The reference's quantised "S" engines are closed (R:thestage_speechkit/nvidia/asr_pipeline.py:47-56,
R:benchmark/README.md:92-98), so until round 5 the fp8 context was held to the repo's own restatement of its own choices
(oracle/whisper_oracle.py: OracleWhisperMXFP8).  This generator produces what those choices MEAN in terms of the reference's model
class: the SAME HF ``WhisperForConditionalGeneration`` (R:thestage_speechkit/nvidia/asr_pipeline.py:57-60) in float32 on CPU, whose
parameters are replaced by exactly what the W8A16 context stores -

  * every decoder ``nn.Linear`` weight (and the tied logits matrix) de-quantised from MXFP8: blocks of 32 along K as the gfx950 scaled
    MFMA groups them, one power-of-two scale per block (OCP MX e8m0, one exponent above the OCP choice so that no element exceeds 256),
    elements cast with **torch** ``float8_e4m3fn`` (the external definition of e4m3 rounding);
  * the pre-LayerNorm of a projection folded into its weight BEFORE quantisation, W' = bf16(g * W), the LayerNorm module left
    identity-affine and its shift moved into the projection's bias (cb = W . beta + b) - algebraically the same model;
  * cross-attention K / V stored as e4m3 with one power-of-two scale per (key, head): forward hooks on ``encoder_attn.{k,v}_proj``;
  * everything else (encoder, embeddings, biases) rounded to bf16, the type the context keeps them in -

teacher-forced along the fp32 greedy path of the golden file (one pass, no cache, as make_golden_ctrl.py does), cross-attention rows
handed to HF's own ``_extract_token_timestamps``.  What the engine adds to THIS model is only its bf16 activation arithmetic, so
tests/test_gpu_full_depth.py holds the fp8a16 context to 1.25 x what HF-bf16 itself loses against HF-fp32 (ctrl_<case>.npz).

Stored under tests/golden/fp8_<case>.npz (clips = the control clips of the case):
  logits_sample [clips, L - 1, V / stride], logits_top [clips, L - 1, 8] (at the fp32 golden's top-8 indices), argmax [clips, L - 1],
  dtw_matrix [clips, new tokens - 1, T], token_timestamps [clips, L], weight_rel_err (mean relative quantisation error of the weights).
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
from .make_golden_ctrl import CTRL, _Outputs  # noqa: E402
from .make_golden_full import PROMPT  # noqa: E402

CASES = ["full_large-v3_c15", "full_large-v3_c15_b4"]


def bf16(t: torch.Tensor) -> torch.Tensor:
    return t.to(torch.bfloat16).to(torch.float32)


def _pow2_scale(amax: torch.Tensor) -> torch.Tensor:
    """2^(sb - 127), sb = max(E - 7, 1), E = biased exponent of the block's largest magnitude (k_decode.hip: sk_quant_mx8)."""
    _, e = torch.frexp(amax.clamp_min(2.0 ** -126))          # amax = m 2^e, m in [0.5, 1): floor(log2 amax) = e - 1
    sb = (e - 1 + 127 - 7).clamp_min(1)
    return torch.ldexp(torch.ones_like(amax), sb - 127)


def mx8_dequant(w: torch.Tensor) -> torch.Tensor:
    """Quantise + de-quantise the last axis (K % 128 == 0) of a bf16-valued tensor: the blocks of 32 the scaled MFMA scales together
    under the engine's operand layout - within each 128-wide step, k = (2h + mm) * 32 + (2u + kk) * 8 + e is block (h, u) - and the
    e4m3 rounding of torch's own float8 type."""
    K = w.shape[-1]
    assert K % 128 == 0
    v = w.reshape(*w.shape[:-1], K // 128, 2, 2, 2, 2, 8)       # [.., s, h, mm, u, kk, e]
    X = _pow2_scale(v.abs().amax(dim=(-4, -2, -1), keepdim=True))
    q = (v / X).to(torch.float8_e4m3fn).to(torch.float32)
    assert torch.isfinite(q).all()
    return (q * X).reshape(w.shape)


def kv8_dequant(x: torch.Tensor, heads: int) -> torch.Tensor:
    """[B, T, d] projection output -> bf16 -> e4m3 with one power-of-two scale per (key, head) -> float32 (k_gemm.hip: gemm_epilogue_kv8)."""
    B, T, d = x.shape
    v = bf16(x).reshape(B, T, heads, d // heads)
    X = _pow2_scale(v.abs().amax(dim=-1, keepdim=True))
    q = (v / X).to(torch.float8_e4m3fn).to(torch.float32)
    return (q * X).reshape(B, T, d)


def fold(ln: torch.nn.LayerNorm, lin: torch.nn.Linear, quantise=True) -> torch.nn.Linear:
    """LayerNorm(g, beta) followed by Linear(W, b)  ==  LayerNorm(1, 0) followed by Linear(W', cb),  W' = bf16(g W), cb = W beta + b;
    W' stored as MXFP8.  Returns the replacement Linear (always with a bias)."""
    g, beta = bf16(ln.weight.data), bf16(ln.bias.data)
    W = bf16(lin.weight.data)
    Wf = bf16(W * g[None, :])
    cb = (W.double() @ beta.double()).float()
    if lin.bias is not None:
        cb = cb + bf16(lin.bias.data)
    new = torch.nn.Linear(W.shape[1], W.shape[0], bias=True)
    new.weight.data = mx8_dequant(Wf) if quantise else Wf
    new.bias.data = cb
    return new


def plain(lin: torch.nn.Linear) -> None:
    lin.weight.data = mx8_dequant(bf16(lin.weight.data))
    if lin.bias is not None:
        lin.bias.data = bf16(lin.bias.data)


def to_w8a16(model, dims) -> float:
    """In-place: the parameters the W8A16 context stores (module docstring).  Returns the mean relative weight error of the decoder."""
    with torch.no_grad():
        for p in model.model.encoder.parameters():
            p.data = bf16(p.data)
        dec = model.model.decoder
        E = bf16(dec.embed_tokens.weight.data).clone()
        errs = []
        for layer in dec.layers:
            ref = [bf16(m.weight.data).clone() for m in (layer.self_attn.q_proj, layer.fc1, layer.fc2)]
            ln = layer.self_attn_layer_norm
            for name in ("q_proj", "k_proj", "v_proj"):
                setattr(layer.self_attn, name, fold(ln, getattr(layer.self_attn, name)))
            layer.encoder_attn.q_proj = fold(layer.encoder_attn_layer_norm, layer.encoder_attn.q_proj)
            layer.fc1 = fold(layer.final_layer_norm, layer.fc1)
            for ln in (layer.self_attn_layer_norm, layer.encoder_attn_layer_norm, layer.final_layer_norm):
                ln.weight.data.fill_(1.0)
                ln.bias.data.zero_()
            for lin in (layer.self_attn.out_proj, layer.encoder_attn.out_proj, layer.fc2):
                plain(lin)
            for lin in (layer.encoder_attn.k_proj, layer.encoder_attn.v_proj):      # bf16 GEMM on the encoder side; the OUTPUT is stored as fp8
                lin.weight.data = bf16(lin.weight.data)
                if lin.bias is not None:
                    lin.bias.data = bf16(lin.bias.data)
                lin.register_forward_hook(lambda m, i, o: kv8_dequant(o, dims.heads))
            errs.append(float((layer.fc2.weight.data - ref[2]).norm() / ref[2].norm()))
        # tied logits matrix: the embedding lookup keeps bf16(E), the projection reads MXFP8 of bf16(g E) with the final LayerNorm folded
        model.proj_out = fold(dec.layer_norm, _lin_of(E))
        dec.layer_norm.weight.data.fill_(1.0)
        dec.layer_norm.bias.data.zero_()
        dec.embed_tokens.weight.data = E
        dec.embed_positions.weight.data = bf16(dec.embed_positions.weight.data)
    return float(np.mean(errs))


def _lin_of(W: torch.Tensor) -> torch.nn.Linear:
    lin = torch.nn.Linear(W.shape[1], W.shape[0], bias=False)
    lin.weight.data = W.clone()
    return lin


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
    model.config._attn_implementation = "eager"
    werr = to_w8a16(model, dims)
    fe = hr.build_feature_extractor(dims, chunk_s)
    pcm = np.stack([wo.synth_audio(16000 * chunk_s, int(z["clip_seeds"][i]), str(z["clip_kinds"][i])) for i in clips])
    mel = fe([p for p in pcm], sampling_rate=16000, return_tensors="pt").input_features
    heads = [tuple(int(x) for x in h) for h in z["alignment_heads"]]
    print(f"[{name}] W8A16 model ready in {time.time() - t0:.0f} s (mean relative fc2 weight error {werr:.4f}); clips {clips}, L = {L}", flush=True)
    t0 = time.time()
    enc = model.model.encoder(mel).last_hidden_state
    dec = model.model.decoder(input_ids=seq[:, :-1], encoder_hidden_states=enc, output_attentions=True, use_cache=False)
    logits = model.proj_out(dec.last_hidden_state).float()
    cross = tuple(dec.cross_attentions)
    assert tuple(cross[0].shape) == (nB, dims.heads, L - 1, T)
    surfaces = []
    orig = gw._dynamic_time_warping

    def spy(matrix):
        surfaces.append(np.array(matrix, dtype=np.float64))
        return orig(matrix)

    gw._dynamic_time_warping = spy
    try:
        ts = model._extract_token_timestamps(_Outputs(sequences=seq, cross_attentions=(cross,)), heads,
                                             num_frames=torch.tensor([2 * T] * nB), num_input_ids=len(PROMPT))
    finally:
        gw._dynamic_time_warping = orig
    out = dict(clips=np.array(clips), weight_rel_err=np.float32(werr),
               versions=np.array([f"transformers {__import__('transformers').__version__}", f"torch {torch.__version__}"]),
               dtw_matrix=-np.stack(surfaces).astype(np.float32), token_timestamps=ts.numpy().astype(np.float32),
               logits_top=torch.gather(logits, 2, top_idx[:, : L - 1]).numpy().astype(np.float32),
               logits_sample=logits[:, :, ::stride].numpy().astype(np.float32), argmax=logits.argmax(dim=-1).numpy().astype(np.int32))
    g = z["dtw_matrix"][clips]
    rel = [float(np.linalg.norm(out["dtw_matrix"][i] - g[i]) / np.linalg.norm(g[i])) for i in range(nB)]
    gl = z["logits_sample"][clips][:, : L - 1]
    lrel = float(np.linalg.norm(out["logits_sample"] - gl) / np.linalg.norm(gl))
    n_flip = int((out["argmax"] != z["logits_top_idx"][clips][:, : L - 1, 0]).sum())
    dev = np.abs(out["token_timestamps"] - z["token_timestamps"][clips])
    print(f"[{name}] HF-fp32 over W8A16 parameters ({time.time() - t0:.0f} s) vs the unquantised fp32 golden: logits rel-L2 {lrel:.4f}, "
          f"top-8 max-abs {float(np.abs(out['logits_top'] - z['logits_top'][clips][:, : L - 1]).max()):.4f}, arg-max differs on {n_flip} of "
          f"{nB * (L - 1)} steps, surface rel-L2 {np.round(rel, 4).tolist()}, timestamps within one frame {float((dev <= 0.0201).mean()):.3f}", flush=True)
    np.savez_compressed(os.path.join(OUT, f"fp8_{name}.npz"), **out)


def main():
    torch.set_grad_enabled(False)
    torch.set_num_threads(os.cpu_count() or 1)
    for name in (sys.argv[1:] or CASES):
        run_case(name)


if __name__ == "__main__":
    main()
