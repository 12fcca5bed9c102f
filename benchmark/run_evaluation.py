#!/usr/bin/env python3
"""WER / CER / RTFx of the MI355X backend on local manifests (counterpart of R:benchmark/run_evaluation.py, which pulls the
Open-ASR datasets and the TensorRT engines from the network).

    python benchmark/run_evaluation.py --model <HF checkpoint dir> --manifest name=path.jsonl [--manifest ...] \
        [--chunk-length-s 20] [--batch-size 32] [--output-dir results]

A manifest line: {"audio": "clip.wav" | "clip.npy", "text": "reference transcript", "language": "en"}; WAV must be 16 kHz
PCM (mono or multi-channel), .npy a float array at 16 kHz.  Generate kwargs as in the reference: greedy, task=transcribe,
max_new_tokens=256 (R:benchmark/run_evaluation.py:94-100).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import wave

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

from eval_utils import evaluate_dataset, mean_over_tasks  # noqa: E402


def load_audio(path: str) -> np.ndarray:
    if path.endswith(".npy"):
        return np.asarray(np.load(path), dtype=np.float32).reshape(-1)
    with wave.open(path, "rb") as wf:
        if wf.getframerate() != 16000 or wf.getsampwidth() != 2:
            raise ValueError(f"{path}: expected 16 kHz 16-bit PCM")
        x = np.frombuffer(wf.readframes(wf.getnframes()), dtype="<i2").astype(np.float32) / 32768.0
        ch = wf.getnchannels()
        return x.reshape(-1, ch).mean(axis=1) if ch > 1 else x


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", required=True)
    ap.add_argument("--manifest", action="append", required=True, help="name=path.jsonl (repeatable: one task per manifest)")
    ap.add_argument("--chunk-length-s", type=int, default=20)
    ap.add_argument("--batch-size", type=int, default=32)
    ap.add_argument("--output-dir", default="results")
    ap.add_argument("--decoder-weights", default=None, choices=[None, "fp8"])
    a = ap.parse_args()
    import torch

    from thewhisper_amd import ASRPipeline

    pipe = ASRPipeline(a.model, chunk_length_s=a.chunk_length_s, device="cuda", torch_dtype=torch.bfloat16,
                       batch_size=a.batch_size, decoder_weights=a.decoder_weights)
    gk = {"num_beams": 1, "task": "transcribe", "do_sample": False, "max_new_tokens": 256}
    results = {}
    for spec in a.manifest:
        name, path = spec.split("=", 1) if "=" in spec else (os.path.basename(spec), spec)
        items = [json.loads(l) for l in open(path) if l.strip()]
        base = os.path.dirname(os.path.abspath(path))
        audio = [load_audio(os.path.join(base, it["audio"])) for it in items]
        lang = items[0].get("language", "en") if items else "en"
        results[name] = evaluate_dataset(lambda x, generate_kwargs: pipe(x, generate_kwargs=generate_kwargs, batch_size=a.batch_size),
                                         audio, [it["text"] for it in items], language=lang, generate_kwargs=gk,
                                         batch_size=a.batch_size)
    results["mean"] = mean_over_tasks(results)
    os.makedirs(a.output_dir, exist_ok=True)
    with open(os.path.join(a.output_dir, "eval_results.json"), "w", encoding="utf-8") as f:
        json.dump(results, f, ensure_ascii=False, indent=2)
    print(json.dumps(results, indent=1))


if __name__ == "__main__":
    main()
