#!/usr/bin/env python3
"""BASELINE config 3 (one 60 s stream, the reference scheduler's 117 rolling-buffer calls, tests/golden/config3_trace.json) through
`AMDWhisperBackend` three ways: plain (what the reference does: every tick decodes the whole buffer), `reuse_committed_prefix` (round 4:
approximate) and `draft_previous_tick` (round 6: exact - tw_greedy_opts::n_draft).  Prints one JSON line per variant.

    python tools/bench_draft.py [--model large-v3] [--dtype bf16] [--calls 117] [--variants plain,force,draft]
Environment: TW_DRAFT_RETRY_ROWS / TW_DRAFT_MAX_ROUNDS / TW_DRAFT_FIRST_ROWS (api.hip) for the verify policy."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch


def main():
    import bench
    from thewhisper_amd import AMDWhisperBackend, ASRPipeline, synthetic
    from thewhisper_amd.engine import WhisperEngine
    from transformers import WhisperFeatureExtractor

    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="large-v3")
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--calls", type=int, default=117)
    ap.add_argument("--variants", default="plain,force,draft")
    ap.add_argument("--dec-layers", type=int, default=0, help="override the decoder depth (0 = the model's)")
    args = ap.parse_args()
    dims = dict(bench.DIMS[args.model])
    if args.dec_layers:
        dims["dec_layers"] = args.dec_layers
    heads = bench.alignment_heads(dims)
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    trace = json.load(open(os.path.join(ROOT, "tests", "golden", "config3_trace.json")))
    chunk_s = trace["chunk_length_s"]
    tdt = {"bf16": torch.bfloat16, "f16": torch.float16, "f32": torch.float32}.get(args.dtype, torch.bfloat16)
    eng = WhisperEngine(dims, 50 * chunk_s, max_batch=1, dtype=args.dtype, alignment_heads=heads, use_graph=True)
    eng.load_state_dict(bench.random_state_dict(dims, torch.device("cuda", 0), seed=0))
    model = synthetic.skeleton_model(dims, device="cuda:0", dtype=tdt, alignment_heads=heads)
    pipe = ASRPipeline(model, feature_extractor=WhisperFeatureExtractor(feature_size=dims["n_mels"], chunk_length=chunk_s),
                       tokenizer=synthetic.build_tokenizer(dims["vocab"]), chunk_length_s=chunk_s, device="cuda:0", torch_dtype=tdt,
                       batch_size=1, engine=eng)
    stream = (np.random.default_rng(trace["seed"]).standard_normal(16000 * trace["seconds"]) * 0.1).clip(-1, 1).astype(np.float32)
    seen = []
    inner = eng.generate_greedy

    def recording(prompt, **kw):
        out = inner(prompt, **kw)
        n0 = 3
        eos = int(kw.get("eos_id", 50257))
        for row in out["sequences"][:, n0:]:
            hit = np.nonzero(row == eos)[0]
            seen.append(np.asarray(row[: int(hit[0])] if len(hit) else row, dtype=np.int64))
        return out

    eng.generate_greedy = recording
    calls = trace["calls"][: args.calls]
    results = {}
    ids_of = {}
    for variant in args.variants.split(","):
        be = AMDWhisperBackend(None, chunk_length_s=chunk_s, asr_pipeline=pipe, reuse_committed_prefix=(variant == "force"),
                               draft_previous_tick=(variant == "draft"))
        be.transcribe(stream[:16000], 0.0, 16000)     # plan learning / graph capture outside the timed calls
        be.reset()
        lat, ids, words = [], [], []
        for c in calls:
            buf = stream[c["offset"] : c["offset"] + c["n"]]
            seen.clear()
            t0 = time.perf_counter()
            w = be.transcribe(buf, c["t0"], 16000)
            lat.append((time.perf_counter() - t0) * 1e3)
            ids.append(np.concatenate(seen) if seen else np.zeros(0, np.int64))
            words.append(w)
        ids_of[variant] = ids
        ls = sorted(lat)
        r = {"variant": variant, "model": args.model, "dtype": args.dtype, "calls": len(lat), "p50_ms": round(ls[len(ls) // 2], 2),
             "p90_ms": round(ls[(len(ls) * 9) // 10], 2), "mean_ms": round(float(np.mean(lat)), 2), "sum_ms": round(sum(lat), 1),
             "tokens": int(sum(len(x) for x in ids))}
        st = be.reuse_stats
        if variant != "plain":
            r.update({k: st[k] for k in st})
            base = ids_of.get("plain")
            if base is not None:
                ident = []
                for a, b in zip(base, ids):
                    n = max(len(a), len(b))
                    if n:
                        m = min(len(a), len(b))
                        ident.append(float((a[:m] == b[:m]).sum()) / n)
                r["token_identity_mean"] = round(float(np.mean(ident)), 4)
                r["calls_identical"] = int(sum(1 for a, b in zip(base, ids) if len(a) == len(b) and (a == b).all()))
        if variant == "draft":
            r["acceptance"] = round(st["confirmed_tokens"] / max(1, st["draft_tokens"]), 3)
        results[variant] = r
        print(json.dumps(r), flush=True)


if __name__ == "__main__":
    main()
