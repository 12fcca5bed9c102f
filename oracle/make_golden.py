"""Generates tests/golden/* by running the REFERENCE itself on CPU in this container.

    python -m oracle.make_golden            (needs /root/reference; never run on the GPU box)

The reference ships no fixtures, so these vectors are produced by importing its own code:
``thestage_speechkit.nvidia.ASRPipeline`` (HF branch, R:thestage_speechkit/nvidia/asr_pipeline.py:57-92) and
``thestage_speechkit.streaming.StreamingPipeline`` with its ``LocalWhisperBackend`` logic
(R:thestage_speechkit/streaming/streaming_pipeline.py:340-435, :443-988), driven with the oracle's seeded
weights + the synthetic tokenizer under the installed transformers 5.15.0.
The weights are regenerated from the seed by ``oracle.whisper_oracle.make_weights`` wherever the
fixtures are consumed, so only small outputs are stored.
"""
from __future__ import annotations

import json
import os
import sys

import numpy as np
import torch

REF = "/root/reference"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tests", "golden")

from . import hf_reference as hr  # noqa: E402
from . import whisper_oracle as wo  # noqa: E402

CASES = [
    # name, preset, chunk_s, audio seconds, audio kind, audio seed, batch_size, max_new_tokens
    ("micro_c10", "micro", 10, 23, "speechlike", 5, 4, 32),
    ("micro80_c30", "micro80", 30, 41, "speechlike", 7, 2, 24),
    ("micro_c10_noise", "micro", 10, 12, "noise", 3, 4, 32),
    # BASELINE configs[0] literally: whisper-tiny.en dimensions, ONE 30 s clip of
    # default_rng(0).standard_normal * 0.1, chunk_length_s = 30, greedy, through the reference's nvidia.ASRPipeline on CPU in fp32,
    # called as R:examples/run_nvidia_asr.py:15-36 calls it (chunk_length_s - 1 at call time)
    ("tiny_en_c30", "tiny.en", 30, 30, "noise", 0, 1, 48),
]
# weights of a case when not make_weights' defaults (stored with the case; the consumers regenerate them from this)
WEIGHT_KW = {
    # unit-gain random weights attend almost uniformly; over 1500 frames the word-timestamp surface is then flat to rounding error and
    # its DTW arg-min a coin toss between implementations (seconds apart).  Sharper queries give a surface with structure.
    "tiny_en_c30": {"scale": 1.0, "q_gain": 6.0},
}


def _import_reference():
    from .ref_bundle import import_reference

    ASRPipeline, sp, _streams = import_reference(REF)
    return ASRPipeline, sp


def build_reference_pipeline(preset: str, chunk_s: int, batch_size: int, weight_seed: int = 0, weight_kw=None):
    ASRPipeline, _ = _import_reference()
    dims = wo.PRESETS[preset]
    w = wo.make_weights(dims, weight_seed, **(weight_kw or {}))
    model = hr.build_hf_model(dims, w)
    pipe = ASRPipeline(model, feature_extractor=hr.build_feature_extractor(dims, chunk_s), tokenizer=hr.build_tokenizer(dims),
                       chunk_length_s=chunk_s, device="cpu", torch_dtype=torch.float32, batch_size=batch_size)
    # version-drift fix D1: 5.x looks positions up through num_embeddings, patch_hf_model only swaps .weight.data
    enc = model.model.encoder
    enc.embed_positions.num_embeddings = enc.embed_positions.weight.shape[0]
    return pipe, dims, w


def main():
    torch.set_grad_enabled(False)
    os.makedirs(OUT, exist_ok=True)
    _, sp = _import_reference()
    summary = {}
    only = set(sys.argv[1:])       # `python -m oracle.make_golden tiny_en_c30`: regenerate the named cases, keep the others
    if only:
        summary = json.load(open(os.path.join(OUT, "pipeline_golden.json")))
    for name, preset, chunk_s, secs, kind, seed, bs, max_new in CASES:
        if only and name not in only:
            continue
        pipe, dims, w = build_reference_pipeline(preset, chunk_s, bs, weight_kw=WEIGHT_KW.get(name))
        audio = wo.synth_audio(16000 * secs, seed, kind)
        gk = {"num_beams": 1, "do_sample": False, "use_cache": True, "language": "en", "max_new_tokens": max_new}
        rec = {"preset": preset, "chunk_s": chunk_s, "seconds": secs, "kind": kind, "seed": seed, "batch_size": bs,
               "max_new_tokens": max_new, "weight_seed": 0, "outputs": {}}
        if name in WEIGHT_KW:
            rec["weight_kw"] = WEIGHT_KW[name]
        for rt in (False, True, "word"):
            out = pipe(audio.copy(), generate_kwargs=dict(gk), chunk_length_s=chunk_s - 1, return_timestamps=rt)
            rec["outputs"][str(rt)] = json.loads(json.dumps(out))  # tuples -> lists
        # low-level vectors from the same HF model: features, encoder output, first-chunk logits
        fe = pipe.feature_extractor
        clip = audio[: chunk_s * 16000]
        feats = fe(clip, sampling_rate=16000, return_tensors="pt").input_features
        enc = pipe.model.model.encoder(feats).last_hidden_state
        ids = torch.tensor([[50258, 50259, 50360, 50364, 100, 2000, 31000]])
        logits = pipe.model(input_features=feats, decoder_input_ids=ids).logits
        np.savez_compressed(
            os.path.join(OUT, f"{name}.npz"),
            mel_rows=feats[0, ::16, ::25].numpy().astype(np.float32),
            mel_mean=np.float32(feats.mean().item()),
            enc_rows=enc[0, ::50, ::8].numpy().astype(np.float32),
            logits_top=torch.topk(logits[0], 8, dim=-1).values.numpy().astype(np.float32),
            logits_top_idx=torch.topk(logits[0], 8, dim=-1).indices.numpy().astype(np.int32),
            teacher_ids=ids.numpy().astype(np.int32),
        )
        summary[name] = rec

    if only:
        with open(os.path.join(OUT, "pipeline_golden.json"), "w") as f:
            json.dump(summary, f, indent=1)
        print("rewrote", sorted(only))
        return
    # streaming: the reference scheduler + its LocalWhisperBackend.transcribe body around the CPU pipeline
    pipe, dims, w = build_reference_pipeline("micro", 10, 1)
    backend = sp.LocalWhisperBackend.__new__(sp.LocalWhisperBackend)
    backend.chunk_length_s = 10
    backend.sample_rate = 16000
    backend.device = "cpu"
    backend.language = "en"
    backend.asr_pipeline = pipe
    calls = []
    orig = backend.transcribe

    def spy(audio, buffer_start_time, sample_rate):
        res = orig(audio, buffer_start_time, sample_rate)
        # where in the source clip this rolling buffer starts (so the call can be replayed without the scheduler)
        cand = np.where(np.isclose(full_audio, audio[0]))[0]
        off = next(int(c) for c in cand if c + len(audio) <= len(full_audio) and np.array_equal(full_audio[c : c + len(audio)], audio))
        calls.append({"n": int(len(audio)), "offset": off, "t0": float(buffer_start_time), "result": res})
        return res

    full_audio = None
    backend.transcribe = spy
    stream = sp.StreamingPipeline(backend=backend, chunk_length_s=10, min_process_chunk_s=0.5, use_vad=False)
    audio = wo.synth_audio(16000 * 14, 11, "speechlike")
    full_audio = audio
    step = 800  # 0.05 s
    committed_all, last_uncommitted = [], []
    for i in range(0, len(audio), step):
        committed, uncommitted = stream(audio[i : i + step])
        committed_all += committed
        last_uncommitted = uncommitted
    summary["streaming_micro_c10"] = {
        "seconds": 14, "seed": 11, "kind": "speechlike", "step_samples": step,
        "calls": json.loads(json.dumps(calls)), "committed": json.loads(json.dumps(committed_all)),
        "uncommitted": json.loads(json.dumps(last_uncommitted)),
    }
    # BASELINE config 3's call pattern from the reference's scheduler + stepper (oracle/config3_trace.py)
    from . import config3_trace
    from .ref_bundle import import_reference

    _, sp3, streams3 = import_reference(REF)
    tr = config3_trace.trace(sp3, streams3)
    with open(os.path.join(OUT, "config3_trace.json"), "w") as f:
        json.dump(tr, f, separators=(",", ":"))
    import transformers

    summary["_meta"] = {"transformers": transformers.__version__, "torch": torch.__version__,
                        "generator": "oracle/make_golden.py", "reference": "TheStageAI/TheWhisper @ /root/reference"}
    with open(os.path.join(OUT, "pipeline_golden.json"), "w") as f:
        json.dump(summary, f, indent=1)
    print("wrote", OUT, {k: (len(v.get("calls", [])) if isinstance(v, dict) else None) for k, v in summary.items()})


if __name__ == "__main__":
    main()
