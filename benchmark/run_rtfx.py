#!/usr/bin/env python3
"""RTFx / TTFT with the reference's definitions (R:benchmark/README.md:92-98, R:benchmark/eval_utils.py:112-154):

    RTFx = audio seconds / transcription wall time on a 10-minute input;   TTFT = start of inference -> first token.

The reference measures its turbo engines on Open-ASR audio with downloaded weights; neither is available offline, so this
leg runs the SAME call - `pipe(audio, batch_size=bs, generate_kwargs=...)` on the ASRPipeline of this backend - on a
synthetic 10-minute clip with random-init weights of the whisper-large-v3-turbo architecture, and forces a fixed number of
new tokens per 30 s window (default 100, about the token rate of conversational English) because a random model's natural
stopping point is arbitrary.  Prints one JSON line; compare with BASELINE.md's turbo rows (other hardware).

    python benchmark/run_rtfx.py [--model large-v3-turbo] [--minutes 10] [--batch-sizes 1,32] [--tokens 100]
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
from transformers import WhisperFeatureExtractor  # noqa: E402

from thewhisper_amd import ASRPipeline, synthetic  # noqa: E402
from thewhisper_amd.engine import WhisperEngine  # noqa: E402


def measure(model_name="large-v3-turbo", minutes=10.0, batch_sizes=(1, 32), chunk_s=30, tokens=100, dtype="bf16", device_index=0):
    """The reference's RTFx / TTFT definitions on this backend's ASRPipeline (module docstring); returns the result dict."""
    if not torch.cuda.is_available():
        raise RuntimeError("needs an MI355X")
    dims = synthetic.DIMS[model_name]
    dev = torch.device("cuda", device_index)
    heads = [tuple(h) for h in synthetic.default_alignment_heads(dims["dec_layers"], dims["heads"])]
    rng = np.random.default_rng(0)
    audio = (rng.standard_normal(int(minutes * 60 * 16000)) * 0.1).clip(-1, 1).astype(np.float32)
    res = {"metric": "RTFx (audio s / wall s) and TTFT, reference definitions", "model": model_name, "audio_s": len(audio) / 16000,
           "chunk_s": chunk_s, "forced_new_tokens_per_window": tokens, "dtype": dtype,
           "data": "synthetic gaussian audio, random-init weights", "runs": []}
    sd = synthetic.random_state_dict(dims, dev, 0)
    for bs in [int(x) for x in batch_sizes]:
        eng = WhisperEngine(dims, 50 * chunk_s, max_batch=bs, dtype=dtype, alignment_heads=heads, device=device_index)
        eng.load_state_dict(sd)
        model = synthetic.skeleton_model(dims, device=f"cuda:{device_index}", dtype=torch.bfloat16, alignment_heads=heads)
        pipe = ASRPipeline(model, feature_extractor=WhisperFeatureExtractor(feature_size=dims["n_mels"], chunk_length=chunk_s),
                           tokenizer=synthetic.build_tokenizer(dims["vocab"]), chunk_length_s=chunk_s, device=f"cuda:{device_index}",
                           torch_dtype=torch.bfloat16, batch_size=bs, engine=eng)
        gk = {"num_beams": 1, "do_sample": False, "use_cache": True, "language": "en", "max_new_tokens": tokens,
              "min_new_tokens": tokens}
        for _ in range(2):   # warm-up: graph capture, and the second call runs the learned short-form plan (model.py)
            pipe(audio[: 16000 * chunk_s * min(bs, 2)].copy(), batch_size=bs, generate_kwargs=dict(gk))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = pipe(audio.copy(), batch_size=bs, generate_kwargs=dict(gk))
        torch.cuda.synchronize()
        wall = time.perf_counter() - t0
        # TTFT at this batch size: log-mel + encoder + cross-K/V + the first decode step for one batch of windows
        pcm = torch.from_numpy(audio[: 16000 * chunk_s]).to(dev).repeat(bs, 1)
        prompt = np.tile(np.array([[50258, 50259, 50360, 50364]], dtype=np.int32), (bs, 1))
        ttft = []
        for _ in range(5):
            torch.cuda.synchronize()
            a = time.perf_counter()
            eng.encode(eng.logmel(pcm))
            eng.cross_kv(bs)
            eng.generate_greedy(prompt, max_new_tokens=1)
            ttft.append(time.perf_counter() - a)
        res["runs"].append({"batch_size": bs, "rtfx": round(len(audio) / 16000 / wall, 2), "wall_s": round(wall, 3),
                            "ttft_s": round(sorted(ttft)[len(ttft) // 2], 4), "text_chars": len(out["text"])})
        eng.close()
        del eng, pipe, model
        torch.cuda.empty_cache()
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="large-v3-turbo")
    ap.add_argument("--minutes", type=float, default=10.0)
    ap.add_argument("--batch-sizes", default="1,32")
    ap.add_argument("--chunk-s", type=int, default=30)
    ap.add_argument("--tokens", type=int, default=100)
    ap.add_argument("--dtype", default="bf16")
    args = ap.parse_args()
    if not torch.cuda.is_available():
        raise SystemExit("needs an MI355X")
    print(json.dumps(measure(args.model, args.minutes, [int(x) for x in args.batch_sizes.split(",")], args.chunk_s, args.tokens,
                             args.dtype)), flush=True)


if __name__ == "__main__":
    main()
