"""Stage-by-stage GPU diagnostics (run on the MI355X box through gpurun; prints one line per check and
writes gpurun_out/diag.json).  Not a pytest file: it never stops at the first failure, so that one
GPU call yields a full picture."""
from __future__ import annotations

import json
import os
import sys
import time
import traceback

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import whisper_oracle as wo  # noqa: E402
from tests.util import PROMPT, clips, dims_variant, make_engine, rel_l2  # noqa: E402

RESULTS = []


def record(name, ok, **info):
    RESULTS.append({"name": name, "ok": bool(ok), **{k: (float(v) if isinstance(v, (np.floating, float)) else v) for k, v in info.items()}})
    print(("PASS " if ok else "FAIL ") + name + " " + json.dumps(RESULTS[-1], default=str)[:400], flush=True)


def guarded(fn):
    def run(*a, **k):
        try:
            return fn(*a, **k)
        except Exception as e:  # noqa: BLE001
            record(fn.__name__ + str(a), False, error=repr(e), tb=traceback.format_exc()[-800:])
    return run


@guarded
def check_logmel(n_mels):
    dims = dims_variant("micro", n_mels=n_mels, enc_layers=0, dec_layers=0)
    w = wo.make_weights(dims, 0)
    eng = make_engine(dims, w, T=500, max_batch=4, dtype="f32")
    pcm = clips(160000, ["noise", "sine", "zeros", "speechlike"])
    ref = wo.log_mel(pcm, n_mels)
    got = eng.logmel(torch.from_numpy(pcm).cuda(), out_dtype=torch.float32).cpu().numpy()
    record(f"logmel n_mels={n_mels}", np.abs(got - ref).max() < 2e-4, maxabs=np.abs(got - ref).max(), per_clip=[float(np.abs(got[i] - ref[i]).max()) for i in range(4)])
    short = pcm[:, :100000]
    got2 = eng.logmel(torch.from_numpy(short).cuda(), n_samples=160000, out_dtype=torch.float32).cpu().numpy()
    ref2 = wo.log_mel(short, n_mels, 160000)
    record(f"logmel padded n_mels={n_mels}", np.abs(got2 - ref2).max() < 2e-4, maxabs=np.abs(got2 - ref2).max())
    eng.close()


@guarded
def check_encoder(preset, T, B, dtype, enc_layers, tol, over=None):
    dims = dims_variant(preset, enc_layers=enc_layers, dec_layers=0, **(over or {}))
    w = wo.make_weights(dims, 1)
    eng = make_engine(dims, w, T=T, max_batch=B, dtype=dtype)
    pcm = clips(T * 320, B)
    mel = wo.log_mel(pcm, dims.n_mels)
    om = wo.OracleWhisper(dims, w, T=T)
    t0 = time.time()
    ref = om.encode(mel)
    t_or = time.time() - t0
    got = eng.encode(torch.from_numpy(mel).cuda(), return_hidden=True).cpu().numpy()
    err = rel_l2(got, ref)
    record(f"encoder {preset} T={T} B={B} {dtype} L={enc_layers}", err < tol and np.isfinite(got).all(), rel_l2=err,
           maxabs=float(np.abs(got - ref).max()), per_clip=[rel_l2(got[i], ref[i]) for i in range(B)], oracle_s=t_or)
    eng.close()


@guarded
def check_decoder(preset, T, B, dtype, dec_layers, tol, over=None, n_steps=6):
    dims = dims_variant(preset, enc_layers=1, dec_layers=dec_layers, **(over or {}))
    w = wo.make_weights(dims, 2)
    heads = [(dec_layers - 1, 0), (dec_layers - 1, 1)] if dec_layers > 0 else []
    eng = make_engine(dims, w, T=T, max_batch=B, dtype=dtype, heads=heads)
    pcm = clips(T * 320, B)
    mel = wo.log_mel(pcm, dims.n_mels)
    om = wo.OracleWhisper(dims, w, T=T)
    enc = om.encode(mel)
    eng.encode(torch.from_numpy(mel).cuda())
    eng.cross_kv(B)
    eng.decoder_reset(B)
    rng = np.random.default_rng(3)
    ids = np.concatenate([np.tile(np.array(PROMPT), (B, 1)), rng.integers(0, 50000, size=(B, n_steps))], axis=1)
    cache = om.new_cache(enc)
    worst = 0.0
    worst_abs = 0.0
    for s in range(ids.shape[1]):
        ref, _ = om.decode(ids[:, s : s + 1], cache)
        got = eng.decode_step(ids[:, s].tolist()).cpu().numpy()
        worst = max(worst, rel_l2(got, ref[:, 0]))
        worst_abs = max(worst_abs, float(np.abs(got - ref[:, 0]).max()))
    record(f"decoder {preset} T={T} B={B} {dtype} L={dec_layers}", worst < tol, rel_l2=worst, maxabs=worst_abs)
    eng.close()


@guarded
def check_greedy(preset, T, B, dtype, max_new, use_graph, min_new=0):
    dims = wo.PRESETS[preset]
    w = wo.make_weights(dims, 0)
    heads = [(dims.dec_layers - 1, 0), (dims.dec_layers - 1, 1)]
    eng = make_engine(dims, w, T=T, max_batch=B, dtype=dtype, heads=heads, use_graph=use_graph)
    pcm = clips(T * 320, B)
    mel_ref = wo.log_mel(pcm, dims.n_mels)
    mel = eng.logmel(torch.from_numpy(pcm).cuda(), out_dtype=torch.float32)
    eng.encode(mel)
    eng.cross_kv(B)
    prompt = np.tile(np.array(PROMPT, dtype=np.int32), (B, 1))
    out = eng.generate_greedy(prompt, max_new_tokens=max_new, min_new_tokens=min_new, timestamps=True, want_alignment=True)
    om = wo.OracleWhisper(dims, w, T=T)
    enc = om.encode(mel_ref)
    opt = wo.GreedyOptions(max_new_tokens=max_new, min_new_tokens=min_new, timestamps=True, alignment_heads=heads)
    ref = wo.greedy_generate(om, enc, prompt, opt)
    same = out["sequences"].shape == ref["sequences"].shape and np.array_equal(out["sequences"], ref["sequences"])
    info = {}
    if not same:
        info["got"] = out["sequences"].tolist()
        info["ref"] = ref["sequences"].tolist()
    record(f"greedy {preset} T={T} B={B} {dtype} graph={use_graph} min_new={min_new}", same, len=out["length"], **info)
    if same:
        L = out["length"]
        al = eng.get_alignment(B, L - 1)
        record(f"alignment rows {preset} B={B}", np.abs(al - ref["cross"]).max() < 1e-4, maxabs=float(np.abs(al - ref["cross"]).max()))
        nf = [2 * T] * B
        ts = eng.token_timestamps(B, 3, L, nf)
        rts = wo.token_timestamps(ref["cross"], 3, nf)
        record(f"token timestamps {preset} B={B}", np.abs(ts - rts).max() <= 0.0201, maxabs=float(np.abs(ts - rts).max()),
               exact=bool(np.array_equal(ts, rts)))
    print("timings", eng.last_timings(), flush=True)
    eng.close()


def main():
    print("device", torch.cuda.get_device_name(0), flush=True)
    check_logmel(128)
    check_logmel(80)
    # conv stem only, then +1 layer, +2 layers; f32 strict then bf16
    check_encoder("micro", 100, 2, "f32", 0, 2e-5)
    check_encoder("micro", 100, 2, "f32", 1, 2e-5)
    check_encoder("micro", 100, 2, "f32", 2, 2e-5)
    check_encoder("micro", 500, 3, "f32", 2, 2e-5)
    check_encoder("micro80", 100, 2, "f32", 2, 2e-5)
    check_encoder("micro", 100, 2, "bf16", 0, 2e-2)
    check_encoder("micro", 100, 2, "bf16", 2, 3e-2)
    check_encoder("micro", 500, 4, "bf16", 2, 3e-2)
    check_encoder("large-v3", 500, 1, "f32", 1, 2e-5)
    check_encoder("large-v3", 500, 1, "bf16", 1, 3e-2)
    check_encoder("large-v3", 500, 4, "bf16", 1, 3e-2)   # exercises the 128x128 tiles
    check_encoder("tiny.en", 1500, 1, "f32", 4, 5e-5)
    # decoder: 0 layers = embed + LN + logits GEMV; then full layers
    check_decoder("micro", 100, 2, "f32", 0, 2e-5)
    check_decoder("micro", 100, 2, "f32", 1, 2e-5)
    check_decoder("micro", 100, 3, "f32", 2, 2e-5)
    check_decoder("micro", 100, 2, "bf16", 2, 3e-2)
    check_decoder("micro", 500, 1, "f32", 2, 2e-5)
    check_decoder("large-v3", 500, 1, "f32", 1, 2e-5)
    check_decoder("large-v3", 500, 2, "bf16", 1, 3e-2)
    check_decoder("tiny.en", 1500, 5, "f32", 4, 5e-5)
    check_decoder("micro", 100, 16, "bf16", 2, 3e-2)
    check_decoder("micro", 100, 8, "f32", 2, 2e-5)
    # greedy + processors + alignment + DTW
    check_greedy("micro", 100, 1, "f32", 24, False)
    check_greedy("micro", 100, 3, "f32", 24, False)
    check_greedy("micro", 500, 2, "f32", 40, False, min_new=40)
    check_greedy("micro", 100, 3, "f32", 24, True)
    check_greedy("micro80", 100, 2, "f32", 24, True)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "diag.json"), "w") as f:
        json.dump(RESULTS, f, indent=1, default=str)
    n_fail = sum(1 for r in RESULTS if not r["ok"])
    print(f"DIAG DONE: {len(RESULTS) - n_fail} pass, {n_fail} fail", flush=True)


if __name__ == "__main__":
    main()
