#!/usr/bin/env python3
"""bench.py - BASELINE.json's headline metric on MI355X: transcription tokens/sec, whisper-large-v3, 10 s chunks.

One "step" = one pass of the whole hot path over one batch of synthetic 16 kHz audio that is already
resident in HBM:  log-mel -> encoder -> cross-K/V -> 128-token greedy decode (timestamp grammar on, as the
reference's streaming backend always runs it, R:thestage_speechkit/streaming/streaming_pipeline.py:395-410)
-> alignment DTW (word timestamps).  Per GPU the workload is `--streams` (default 16) concurrent 10 s
chunks = configs[3]'s per-GPU share (128 streams / 8 GPUs); N GPUs run N independent replicas on
disjoint streams (weak scaling, no data-path collective - SURVEY.md section 8e).

Launch: `python bench.py [--gpus N]` - with N > 1 and no WORLD_SIZE in the environment the script re-executes itself as
N ranks (`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ...`, one rank per GPU,
RCCL for the barrier / max / sum); launched by the driver through torch.distributed.run it reads RANK / LOCAL_RANK /
WORLD_SIZE from the environment.  Rank 0 prints ONE JSON line.

`--scaling strong --total-streams 128` (round 5): the streams are a FIXED total sharded over the ranks
(stream i on rank i % N), each rank takes its share in passes of at most 64 streams; the default is weak scaling (--streams per GPU).

Besides the engine-level headline the line carries (rank 0, N = 1 only):
  * the CONTRACT's definitions first: `value_contract` (tokens over wall-clock from transcribe() entry to return,
    host buffers in, words out, 16 free-running sessions behind the BatchingHub), `p50_chunk_latency_contract_ms` (one
    `AMDWhisperBackend.transcribe` call on a 10 s host buffer, R:thestage_speechkit/streaming/streaming_pipeline.py:388-435),
    `hub_request_p50_ms` / `_p90_ms`; the engine-level figures stay beside them (`value`, `p50_chunk_latency_ms`);
  * `config3`: BASELINE config 3's call pattern - the 117 backend calls the REFERENCE'S OWN scheduler + stepper make for one 60 s stream
    (tests/golden/config3_trace.json, generated and re-derived from the reference: oracle/config3_trace.py) - replayed through the
    backend at large-v3 dimensions: p50 / p90 per call, calls per audio second, with and without `reuse_committed_prefix`;
  * `dtype` = f16 since round 6 (the reference's streaming default dtype and the 16-bit context whose greedy ids equal the fp32
    reference's on every full-depth clip); `value_bf16`, `roofline_bf16`: the bfloat16 context on the same schedule; `parity_full_depth`:
    the id-identity figures of both from the committed GPU-suite log (refused when that log belongs to other kernel sources);
  * `pipeline`: everything measured through the drop-in API (hub legs, two cohorts, short passes, lock-step);
  * `cpu_baseline`: the reference's own `nvidia.ASRPipeline` on this box's host cores (bounded sample; `kind: "reference"` wherever the
    reference package is importable - /root/reference, or the oracle/_ref bundle on the GPU box);
  * `roofline`: decode step, algorithmic bytes (W = 2*(Ld*14*d^2 + V*d)) over HIP-event time.
"""
from __future__ import annotations

import argparse
import glob
import importlib
import json
import os
import socket
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

from thewhisper_amd import synthetic  # noqa: E402

DIMS = synthetic.DIMS
random_state_dict = synthetic.random_state_dict
HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: 8.0 TB/s spec
HBM_COPY_CEILING_GBS = 6290.0  # ... 6.29 TB/s measured float4 copy


def alignment_heads(dims):
    return [tuple(h) for h in synthetic.default_alignment_heads(dims["dec_layers"], dims["heads"])]


def algorithmic_decode_bytes(dims, B, T, n_prompt, steps, esz=2, wsz=None):
    """bytes/step = W + B*163840*(T + t) (large-v3 numbers generalised).
    W = esz*(Ld*14*d^2 + V*d) for ffn = 4d: per decoder layer the self q/k/v/out (4 d^2), the cross q/out (2 d^2 - the
    cross K/V projection weights are NOT read during decode) and fc1 + fc2 (2*d*ffn); plus the tied logits matrix.
    Per stream and step the cross K/V (2*Ld*T*d*esz) and the self K/V read so far."""
    d, Ld, V = dims["d_model"], dims["dec_layers"], dims["vocab"]
    wsz = esz if wsz is None else wsz  # bytes per projection weight (MXFP8: 1 + 1/32 for the block scales)
    W = wsz * (Ld * (6 * d * d + 2 * d * dims["ffn"]) + V * d)
    total = 0
    for s in range(steps):
        t = s + 1
        total += W + B * (2 * Ld * d * esz) * (T + t)
    return total, W


def _host_cores() -> int:
    """Cores this process may really use: scheduler affinity capped by the cgroup CPU quota (containers often expose the
    host's core count through os.cpu_count(); running hundreds of threads on a few cores only thrashes)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = max(1, min(n, int(float(quota) / float(period) + 0.5)))
    except Exception:  # noqa: BLE001
        pass
    return max(1, min(n, 64))


# ------------------------------------------------------------------------------------------------------------------
# CPU baseline (reported, not the target): the reference's own pipeline class on the host cores, bounded sample
# ------------------------------------------------------------------------------------------------------------------
def _cpu_baseline_worker(model_name, chunk_s, new_tokens, calls):
    """Child process.  BASELINE.md section 3: `thestage_speechkit.nvidia.ASRPipeline` (HF branch, model_size=None,
    R:thestage_speechkit/nvidia/asr_pipeline.py:57-60) with device="cpu", fp32, greedy, word timestamps on as the reference's
    streaming backend runs it (R:thestage_speechkit/streaming/streaming_pipeline.py:395-410), `calls` calls x `new_tokens`
    forced tokens.  Where /root/reference is absent (the GPU box) the class it subclasses without changing any arithmetic -
    transformers' AutomaticSpeechRecognitionPipeline - is called the same way; the JSON says which one ran.  (Since round 5 the GPU
    box has the reference's package too: oracle/ref_bundle.py.)"""
    from oracle import hf_reference as hr
    from oracle import whisper_oracle as wo
    from transformers import WhisperForConditionalGeneration
    from transformers.initialization import no_init_weights

    cores = _host_cores()
    torch.set_num_threads(cores)
    torch.set_grad_enabled(False)
    dims = wo.PRESETS[model_name]
    t0 = time.time()
    with no_init_weights():
        model = WhisperForConditionalGeneration(hr.build_hf_config(dims))
    g = torch.Generator().manual_seed(0)
    for name, p in model.named_parameters():   # timing does not depend on the values; random so that decoding is not degenerate
        if p.dim() >= 2:
            p.uniform_(-1.7 / p.shape[-1] ** 0.5, 1.7 / p.shape[-1] ** 0.5, generator=g)
        elif "layer_norm" in name and name.endswith("weight"):
            p.fill_(1.0)
        else:
            p.zero_()
    model.eval()
    hr.fill_generation_config(model.generation_config, dims)
    fe, tok = hr.build_feature_extractor(dims, chunk_s), hr.build_tokenizer(dims)
    from oracle import ref_bundle   # /root/reference here; on the GPU box the package __graft_entry__.build() packed into oracle/_ref/

    if ref_bundle.reference_dir() is not None:
        ASRPipeline, _sp, _streams = ref_bundle.import_reference()   # the reference's own class (importing it also installs its LCS patch)
        pipe = ASRPipeline(model, feature_extractor=fe, tokenizer=tok, chunk_length_s=chunk_s, device="cpu",
                           torch_dtype=torch.float32, batch_size=1)
        enc = model.model.encoder   # version-drift fix D1: 5.x looks positions up through num_embeddings
        enc.embed_positions.num_embeddings = enc.embed_positions.weight.shape[0]
        which = f"thestage_speechkit.nvidia.ASRPipeline (the reference's class, HF branch; imported from {ref_bundle.which()})"
    else:
        from transformers import AutomaticSpeechRecognitionPipeline

        from thewhisper_amd import lcs_patch  # noqa: F401  the reference's chunk-merge fix, as importing thestage_speechkit installs it

        hr.patch_chunk_length(model, chunk_s)
        pipe = AutomaticSpeechRecognitionPipeline(model, feature_extractor=fe, tokenizer=tok, device="cpu", chunk_length_s=chunk_s,
                                                  torch_dtype=torch.float32, batch_size=1)
        which = ("transformers AutomaticSpeechRecognitionPipeline + patch_hf_model arithmetic (what thestage_speechkit.nvidia."
                 "ASRPipeline is a constructor around; /root/reference is absent on this box)")
    t_init = time.time() - t0
    gk = {"use_cache": True, "num_beams": 1, "do_sample": False, "language": "en", "max_new_tokens": new_tokens,
          "min_new_tokens": new_tokens}
    fe(wo.synth_audio(16000, 0, "noise"), sampling_rate=16000, return_tensors="pt")   # untimed: first-call cost of torch.stft
    dts = []
    for i in range(calls):
        pcm = wo.synth_audio(chunk_s * 16000, i, "noise")
        t0 = time.time()
        pipe(pcm, return_timestamps="word", generate_kwargs=dict(gk), chunk_length_s=chunk_s)
        dts.append(time.time() - t0)
    print("CPU_BASELINE " + json.dumps({"tok": new_tokens * calls, "dt": sum(dts), "calls": dts, "init": t_init, "cores": cores,
                                        "which": which}), flush=True)


def _reference_available() -> bool:
    try:
        from oracle import ref_bundle

        return ref_bundle.reference_dir() is not None
    except Exception:  # noqa: BLE001
        return False


def cpu_baseline(model_name, chunk_s, new_tokens, calls, budget_s=240):
    """Bounded sample in a child process with a hard wall-clock budget.  oracle/ is imported only here."""
    cores = _host_cores()
    code = f"import bench; bench._cpu_baseline_worker({model_name!r}, {chunk_s}, {new_tokens}, {calls})"
    env = dict(os.environ, HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES="")
    what = (f"{model_name} dims (random init), fp32, {cores} threads, 1 stream, {calls} pipeline calls on a {chunk_s} s chunk, each: "
            f"feature extraction + encoder + {new_tokens} forced greedy tokens + word-timestamp DTW + merge")
    try:
        r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=env, capture_output=True, text=True, timeout=budget_s)
        line = [l for l in r.stdout.splitlines() if l.startswith("CPU_BASELINE ")]
        if not line:
            raise RuntimeError((r.stderr or r.stdout)[-300:])
        d = json.loads(line[-1][len("CPU_BASELINE "):])
        # "reference" only when the reference tree's own class ran; on the GPU box (no /root/reference) it is the class that one
        # subclasses without touching its arithmetic: HF's pipeline, labelled as such
        kind = "reference" if d["which"].startswith("thestage_speechkit") else "hf-pipeline"
        return {"value": round(d["tok"] / d["dt"], 3), "unit": "tok/s", "cores": d["cores"], "kind": kind,
                "sample": f"{d['which']}: {what}; {d['dt']:.1f} s wall ({', '.join(f'{x:.1f}' for x in d['calls'])} s per call; "
                          f"+{d['init']:.0f} s untimed model init)"}
    except subprocess.TimeoutExpired:
        return {"value": round(new_tokens * calls / budget_s, 3), "unit": "tok/s", "cores": cores,
                "kind": "reference" if _reference_available() else "hf-pipeline",
                "sample": what + f": did NOT finish within the {budget_s} s budget - value is an upper bound"}


# ------------------------------------------------------------------------------------------------------------------
# the same workload through the drop-in API (rank 0, N = 1)
# ------------------------------------------------------------------------------------------------------------------
def pipeline_leg(eng, dims, args, device_index, heads, latency_calls, hub_rounds):
    """tokens/s of `sessions` threads calling the reference's `TranscriptionBackend.transcribe` contract through the
    BatchingHub, and p50/p90 wall time of a single `AMDWhisperBackend.transcribe` call on a host float32 buffer (PCIe upload,
    HF chunking, feature extraction, generate control flow, tokenizer decode and LCS merge included)."""
    from transformers import WhisperFeatureExtractor

    from thewhisper_amd import AMDWhisperBackend, ASRPipeline
    from thewhisper_amd.serving import BatchingHub

    B = args.streams
    tdt = {"f16": torch.float16, "f32": torch.float32}.get(args.dtype, torch.bfloat16)   # (the engine holds the arithmetic; this is the HF shell's dtype)
    model = synthetic.skeleton_model(dims, device=f"cuda:{device_index}", dtype=tdt, alignment_heads=heads)
    pipe = ASRPipeline(model, feature_extractor=WhisperFeatureExtractor(feature_size=dims["n_mels"], chunk_length=args.chunk_s),
                       tokenizer=synthetic.build_tokenizer(dims["vocab"]), chunk_length_s=args.chunk_s, device=f"cuda:{device_index}",
                       torch_dtype=tdt, batch_size=B, engine=eng)
    # (the plain backend - every call decoded from scratch, what the reference does; the default draft mode is measured below)
    backend = AMDWhisperBackend(None, chunk_length_s=args.chunk_s, asr_pipeline=pipe, draft_previous_tick=False)
    counted = {"tok": 0}
    inner = eng.generate_greedy

    def counting(prompt, **kw):
        out = inner(prompt, **kw)
        seq, n0 = out["sequences"], np.asarray(prompt).shape[1]
        eos = int(kw.get("eos_id", 50257))
        for row in seq[:, n0:]:
            hit = np.nonzero(row == eos)[0]
            counted["tok"] += int(hit[0]) + 1 if len(hit) else len(row)
        return out

    eng.generate_greedy = counting
    try:
        rng = np.random.default_rng(7)
        clips = [(rng.standard_normal(args.chunk_s * 16000) * 0.1).clip(-1, 1).astype(np.float32) for _ in range(B)]
        for i in range(3):
            backend.transcribe(clips[i % B], 0.0, 16000)
        lat = []
        for i in range(latency_calls):
            t0 = time.perf_counter()
            backend.transcribe(clips[i % B], 0.0, 16000)
            lat.append((time.perf_counter() - t0) * 1e3)
        lat.sort()
        # BASELINE config 3: the call pattern of ONE 60 s stream as the REFERENCE'S OWN scheduler + stepper produce it
        # (tests/golden/config3_trace.json: reference StreamingPipeline(chunk_length_s=10, min_process_chunk_s=0.5, use_vad=False) fed by
        # the reference ArrayStream(step_size_s=0.05, real_time=False), R:thestage_speechkit/streaming/streaming_pipeline.py:740-822,
        # R:thestage_speechkit/streaming/streams.py:16-81; generated by oracle/make_golden.py, re-derived from the reference by
        # tests/test_config3_trace.py; rounds 1-4 replayed a hand-written list of buffer lengths here).  Per call: which samples of
        # the stream the rolling buffer holds and the buffer_start_time the scheduler passed; the audio is SURVEY 8d's seed-0 stream.
        trace = json.load(open(os.path.join(ROOT, "tests", "golden", "config3_trace.json")))
        stream = (np.random.default_rng(trace["seed"]).standard_normal(16000 * trace["seconds"]) * 0.1).clip(-1, 1).astype(np.float32)
        n_init = 3   # decoder prompt (sot, language, task): ids after it are what a call produced
        seen_ids = []

        def run_pattern(be):
            lat_ms, ids = [], []
            if getattr(be, "reuse_committed_prefix", False) or getattr(be, "draft_previous_tick", False):
                # plan learning (one ordinary call on 1 s of silence, streaming.JobCodec.learn) outside the replayed calls: its tokens
                # are not the trace's and its time is paid once per backend, not per tick
                be.transcribe(np.zeros(16000, dtype=np.float32), 0.0, 16000)
                be.reset()
            for c in (trace["calls"] if latency_calls > 0 and args.chunk_s == trace["chunk_length_s"] else []):
                buf = stream[c["offset"] : c["offset"] + c["n"]]
                seen_ids.clear()
                t0 = time.perf_counter()
                be.transcribe(buf, c["t0"], 16000)
                lat_ms.append((time.perf_counter() - t0) * 1e3)
                ids.append(np.concatenate(seen_ids) if seen_ids else np.zeros(0, np.int64))   # every seek pass of the call, eos-trimmed
            return lat_ms, ids

        def counting_ids(prompt, **kw):   # on top of `counting`: the ids a call decoded, BEFORE tokenizer / gibberish filter / word merge
            out = counting(prompt, **kw)
            eos = int(kw.get("eos_id", 50257))
            for row in out["sequences"][:, n_init:]:
                hit = np.nonzero(row == eos)[0]
                seen_ids.append(np.asarray(row[: int(hit[0])] if len(hit) else row, dtype=np.int64))
            return out

        eng.generate_greedy = counting_ids
        rag, plain_ids = run_pattern(backend)
        rag_sorted = sorted(rag)
        reuse_out = {}
        config3 = None
        if rag:
            audio_s = trace["seconds"]
            config3 = {
                "definition": trace["definition"], "trace": "tests/golden/config3_trace.json", "stream_seconds": audio_s, "calls": len(rag),
                "calls_per_audio_s": round(len(rag) / audio_s, 3),
                "p50_ms": round(rag_sorted[len(rag) // 2], 2), "p90_ms": round(rag_sorted[min(len(rag) - 1, (len(rag) * 9) // 10)], 2),
                "max_ms": round(rag_sorted[-1], 2), "sum_ms": round(sum(rag), 1),
                # the whole stream's backend time over its duration: < 1 = the stream is served faster than it is spoken
                "backend_busy_per_audio_s": round(sum(rag) * 1e-3 / audio_s, 4),
                "buffer_seconds_p50": round(float(np.median([c["n"] for c in trace["calls"]])) / 16000, 2),
                "model": f"whisper-{args.model} dims, random weights, {args.dtype}; AMDWhisperBackend.transcribe on host float32 buffers",
            }
        if latency_calls > 0 and rag and args.chunk_s <= 10:
            rb = AMDWhisperBackend(None, chunk_length_s=args.chunk_s, asr_pipeline=pipe, reuse_committed_prefix=True)
            rl, reuse_ids = run_pattern(rb)
            rl.sort()
            ident = []
            for a, b in zip(plain_ids, reuse_ids):
                n = max(len(a), len(b))
                if n:
                    m = min(len(a), len(b))
                    ident.append(float((a[:m] == b[:m]).sum()) / n)
            st = rb.reuse_stats
            reuse_out = {
                "reuse_scheduler_pattern_p50_ms": round(rl[len(rl) // 2], 2), "reuse_scheduler_pattern_p90_ms": round(rl[(len(rl) * 9) // 10], 2),
                "reuse_calls": st["calls"], "reuse_calls_with_forced_prefix": st["reused"],
                "reuse_forced_token_share": round(st["forced_tokens"] / max(1, st["forced_tokens"] + st["decoded_tokens"]), 3),
                "reuse_token_identity_mean": round(float(np.mean(ident)), 3) if ident else None,
                "reuse_note": "AMDWhisperBackend(reuse_committed_prefix=True), opt-in: the previous tick's tokens whose word timestamps end >= 1 s "
                              "before the old buffer's end are forced through a batched prefill (tw_greedy_opts::n_forced), only the tail "
                              "is decoded; token identity = share of the decoded token ids (all seek passes of a call, before the tokenizer and the "
                              "reference's gibberish filter - which empties every large-v3 random-weight transcript) equal, position by position, to "
                              "the plain backend's on the same call - random weights, so a chaotic lower bound",
            }
            if config3 is not None:
                config3["with_reuse_committed_prefix"] = {"p50_ms": reuse_out["reuse_scheduler_pattern_p50_ms"], "p90_ms": reuse_out["reuse_scheduler_pattern_p90_ms"],
                                                          "calls_with_forced_prefix": st["reused"], "token_identity_mean": reuse_out["reuse_token_identity_mean"]}
        # ... and the EXACT form (round 6): `draft_previous_tick` - the previous tick's tokens as a draft the engine verifies
        # (tw_greedy_opts::n_draft); ids must equal the plain backend's on every call
        if latency_calls > 0 and rag:
            db = AMDWhisperBackend(None, chunk_length_s=args.chunk_s, asr_pipeline=pipe, draft_previous_tick=True)
            dl, draft_ids = run_pattern(db)
            dls = sorted(dl)
            ident = []
            for a, b in zip(plain_ids, draft_ids):
                n = max(len(a), len(b))
                if n:
                    m = min(len(a), len(b))
                    ident.append(float((a[:m] == b[:m]).sum()) / n)
            st = db.reuse_stats
            draft_out = {
                "p50_ms": round(dls[len(dls) // 2], 2), "p90_ms": round(dls[(len(dls) * 9) // 10], 2), "sum_ms": round(sum(dl), 1),
                "calls_with_draft": st["reused"], "draft_tokens": st["draft_tokens"], "confirmed_tokens": st["confirmed_tokens"],
                "acceptance": round(st["confirmed_tokens"] / max(1, st["draft_tokens"]), 3), "verify_launches": st["verify_launches"],
                "token_identity_mean": round(float(np.mean(ident)), 4) if ident else None,
                "calls_identical": int(sum(1 for a, b in zip(plain_ids, draft_ids) if len(a) == len(b) and (a == b).all())), "calls": len(dl),
                "note": "AMDWhisperBackend(draft_previous_tick=True): the previous tick's tokens are offered as a draft and VERIFIED (logits + "
                        "processors of every position in batched launches, tw_greedy_opts::n_draft): ids identical to the plain backend by "
                        "construction; acceptance on random weights is a lower bound (the transcript of a random-weight model changes with "
                        "every 0.5 s of new audio)",
            }
            reuse_out["draft_previous_tick"] = draft_out
            if config3 is not None:
                config3["with_draft_previous_tick"] = draft_out
        eng.generate_greedy = counting
        # (a) lock-step rounds (continuity with rounds 1-2: all sessions ask at once and wait for the slowest), classic whole-call
        #     batches: what the hub did until round 3
        def lockstep(hub, rounds):
            gate = threading.Barrier(B)

            def session(k):
                be = hub.stream_backend()
                for _ in range(rounds):
                    gate.wait()
                    be.transcribe(clips[k], 0.0, 16000)

            th = [threading.Thread(target=session, args=(k,)) for k in range(B)]
            [t.start() for t in th]
            [t.join() for t in th]

        # (b) free-running sessions (what B independent streaming schedulers are): every session asks again as soon as it has
        #     its answer; the hub fills each pass with whatever chunks need one (serving.py)
        first_done = {}

        def free_running(hub, per_session, sessions=B):
            first_done.clear()

            def session(k):
                be = hub.stream_backend()
                for i in range(per_session):
                    be.transcribe(clips[(k + i) % B], 0.0, 16000)
                # the first session to finish closes the steady-state window: until here every session had a request in flight
                first_done.setdefault("t", (time.perf_counter(), counted["tok"]))

            th = [threading.Thread(target=session, args=(k,)) for k in range(sessions)]
            [t.start() for t in th]
            [t.join() for t in th]

        out = {}

        def measure(prefix, sessions=B, rounds=hub_rounds, **hub_kw):
            hub = BatchingHub(backend, max_batch=B, max_wait_s=0.004, **hub_kw)
            free_running(hub, 1, sessions)             # warm-up: plan learning + graph capture for this batch size
            hub.latencies.clear(); hub.batches.clear(); counted["tok"] = 0
            p0, r0, f0 = hub.passes, hub.rows, hub.prefetched
            ph0 = dict(hub.phase_s)
            t0 = time.perf_counter()
            free_running(hub, rounds, sessions)
            dt = time.perf_counter() - t0
            lats = sorted(hub.latencies)
            passes, rows = hub.passes - p0, hub.rows - r0
            t1, tok1 = first_done.get("t", (t0 + dt, counted["tok"]))
            res = {
                f"{prefix}tok_per_s": round(counted["tok"] / dt, 1), f"{prefix}sessions": sessions, f"{prefix}requests": sessions * rounds,
                # the same while EVERY session still has requests to send (window closed by the first session that finishes): the
                # whole-run figure above also contains the passes that drain the last requests with fewer and fewer rows
                f"{prefix}steady_tok_per_s": round(tok1 / max(1e-9, t1 - t0), 1),
                f"{prefix}mode": "continuous (seek passes of chunks)" if hub._codec is not None else "whole-call batches",
                f"{prefix}request_p50_ms": round(lats[len(lats) // 2] * 1e3, 2) if lats else None,
                f"{prefix}request_p90_ms": round(lats[min(len(lats) - 1, (len(lats) * 9) // 10)] * 1e3, 2) if lats else None,
                f"{prefix}mean_rows_per_pass": round(rows / max(1, passes), 2), f"{prefix}passes": passes,
                f"{prefix}passes_per_request": round(rows / max(1, sessions * rounds), 2),
                f"{prefix}rows_prefetched": hub.prefetched - f0, f"{prefix}tokens": counted["tok"],
                # batcher-thread wall time per pass by phase (ms): `greedy` = the engine call inside `run`, the rest of `run` is host work
                f"{prefix}phase_ms_per_pass": {k: round((v - ph0.get(k, 0.0)) / max(1, passes) * 1e3, 2) for k, v in hub.phase_s.items()},
            }
            hub.close()
            return res

        # (b1) the headline API reading (round 3's schedule: the encoder stage of a pass runs before its decode loop)
        out.update(measure("hub_"))
        # (b2) the same with arrivals encoded on a sibling context / CU-masked stream while a pass decodes (serving._Prefetcher;
        #      round 4).  Closed-loop sessions come back within milliseconds of a pass's END, i.e. during the intake of the next
        #      pass, not during its decode loop: few rows take the prefetch route (17 of 624) and the extra thread costs more than
        #      they save (9 505 vs 9 739 tok/s, profiles/r04_hub_prefetch_ab.json) - reported, not the default.
        if args.hub_prefetch_cus > 0:
            r = measure("hub_prefetch_", prefetch_cus=args.hub_prefetch_cus)
            out.update({k: r[k] for k in ("hub_prefetch_tok_per_s", "hub_prefetch_request_p50_ms", "hub_prefetch_request_p90_ms",
                                          "hub_prefetch_rows_prefetched")})
        # (b2') two cohorts: 2 x B sessions on the same B-row passes.  The rows that sit a pass out - new arrivals AND chunks that need a
        #      further seek iteration - are encoded on the side stream under the running pass's decode loop and adopted by the next
        #      pass (serving.py: _Prefetcher.ahead): the encoder stage and the sessions' round trips leave the critical path, at twice
        #      the request latency.  `hub_2x_*` with the side stream, `hub_2x_serial_*` without (passes encode their own rows).
        if args.hub_two_cohorts and args.hub_prefetch_cus > 0:
            keys = ("tok_per_s", "steady_tok_per_s", "sessions", "requests", "request_p50_ms", "request_p90_ms", "phase_ms_per_pass", "mean_rows_per_pass", "passes", "rows_prefetched")
            rounds2 = max(2, hub_rounds // 2)
            r = measure("hub_2x_", sessions=2 * B, rounds=rounds2, prefetch_cus=args.hub_prefetch_cus)
            out.update({f"hub_2x_{k}": r[f"hub_2x_{k}"] for k in keys})
            r = measure("hub_2x_serial_", sessions=2 * B, rounds=rounds2)
            out.update({f"hub_2x_serial_{k}": r[f"hub_2x_serial_{k}"] for k in keys[:7]})
        # (b3) short passes: `hub_short_tokens` new tokens per pass instead of 128.  A random-weight decoder closes timestamp pairs at
        #      random places, so a 10 s buffer needs 3-4 seek passes whatever the budget (a trained model: 1-2); with 24-token passes
        #      a request's wall time is what a trained model's ONE 128-token pass costs, which makes the reference scheduler's 0.5 s
        #      cadence testable at 16 sessions per GPU (passes per request are reported with it)
        if args.hub_short_tokens > 0:
            full_kwargs = backend._generate_kwargs
            backend._generate_kwargs = lambda: {**full_kwargs(), "max_new_tokens": args.hub_short_tokens}
            try:
                out.update(measure("hub_short_"))
                out["hub_short_max_new_tokens"] = args.hub_short_tokens
            finally:
                backend._generate_kwargs = full_kwargs
        hub = BatchingHub(backend, max_batch=B, max_wait_s=0.05, continuous=False)
        lockstep(hub, 1)
        counted["tok"] = 0
        t0 = time.perf_counter()
        lockstep(hub, hub_rounds)
        dt = time.perf_counter() - t0
        hub.close()
        out.update({"hub_lockstep_tok_per_s": round(counted["tok"] / dt, 1), "hub_lockstep_ms_per_round": round(dt / hub_rounds * 1e3, 2)})
        out.update({
            "backend_transcribe_p50_ms": round(lat[len(lat) // 2], 2) if lat else None,
            "backend_transcribe_p90_ms": round(lat[min(len(lat) - 1, (len(lat) * 9) // 10)], 2) if lat else None,
            "backend_transcribe_calls": len(lat),
            "scheduler_pattern_p50_ms": round(rag_sorted[len(rag) // 2], 2) if rag else None,
            "scheduler_pattern_p90_ms": round(rag_sorted[min(len(rag) - 1, (len(rag) * 9) // 10)], 2) if rag else None,
            "scheduler_pattern_calls": len(rag),
            "config3": config3,
            **reuse_out,
            "note": f"host float32 {args.chunk_s} s buffers through thewhisper_amd.AMDWhisperBackend.transcribe (reference contract "
                    f"R:thestage_speechkit/streaming/streaming_pipeline.py:388-435: word timestamps on, max_new_tokens=128, natural eos); "
                    f"hub_*: {B} free-running session threads share one engine through BatchingHub, {hub_rounds} requests each, tokens "
                    f"counted over every seek pass (a random-weight model needs ~3 passes per 10 s buffer); hub_lockstep_*: all "
                    f"sessions ask at once and the round ends with the slowest (whole-call batches, as in rounds 1-2); "
                    f"scheduler_pattern_* / config3 = the rolling buffers the reference scheduler sends for one 60 s stream (committed trace)",
        })
        return out
    finally:
        eng.generate_greedy = inner



# ------------------------------------------------------------------------------------------------------------------
# BASELINE.json's other single-GPU configurations, compact, in the same run (rank 0, N = 1)
# ------------------------------------------------------------------------------------------------------------------
def secondary_leg(label, model, chunk_s, B, dtype, steps, new_tokens, device_index, use_graph=True):
    """One more configuration through the same C ABI, stages back to back on the whole chip (no encoder overlap): tokens/s of
    the whole hot path, the decode step and its roofline fraction (same algorithmic-bytes rule as the headline)."""
    from thewhisper_amd.engine import WhisperEngine

    dims = DIMS[model]
    T = 50 * chunk_s
    heads = alignment_heads(dims)
    dev = torch.device("cuda", device_index)
    eng = WhisperEngine(dims, T, max_batch=B, dtype=dtype, alignment_heads=heads, device=device_index, use_graph=use_graph)
    try:
        eng.load_state_dict(random_state_dict(dims, dev, seed=0))
        torch.cuda.empty_cache()
        g = torch.Generator(device=dev)
        g.manual_seed(2000)
        pcm = (torch.randn((B, chunk_s * 16000), device=dev, generator=g, dtype=torch.float32) * 0.1).clamp_(-1, 1)
        prompt = np.tile(np.array([[50258, 50259, 50360]], dtype=np.int32), (B, 1))

        def one():
            eng.encode(eng.logmel(pcm))
            eng.cross_kv(B)
            out = eng.generate_greedy(prompt, max_new_tokens=new_tokens, min_new_tokens=new_tokens, timestamps=True, want_alignment=True)
            eng.token_timestamps(B, 3, out["length"], [2 * T] * B)
            return out["length"] - 3, eng.last_timings()

        one()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        tok, greedy_ms, enc_ms, dsteps = 0, 0.0, 0.0, 0
        for _ in range(steps):
            n, tm = one()
            tok += n * B
            greedy_ms += tm["greedy_ms"]
            enc_ms += tm["encode_ms"] + tm["cross_kv_ms"] + tm["logmel_ms"]
            dsteps += tm["decode_steps"]
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        esz = 2
        wsz = 1.0 + 1.0 / 32 if dtype.startswith("fp8") else None
        spc = dsteps // steps
        alg, _W = algorithmic_decode_bytes(dims, B, T, 3, spc, esz, wsz)
        ach = alg / (greedy_ms / steps * 1e-3) / 1e9
        return {"config": label, "workload": f"whisper-{model}, {chunk_s} s chunk(s), {B} stream(s), {new_tokens} forced tokens + DTW, dtype {dtype}, "
                                            f"stages back to back on the whole chip",
                "tok_per_s": round(tok / dt, 1), "ms_per_step": round(dt / steps * 1e3, 2), "steps": steps,
                "encoder_stage_ms": round(enc_ms / steps, 2), "decode_step_ms": round(greedy_ms / max(1, dsteps), 4),
                "roofline_frac": round(ach / HBM_PEAK_GBS, 4), "achieved_GBs": round(ach, 1),
                "algorithmic_bytes_per_step": int(alg / max(1, spc))}
    finally:
        eng.close()


# ------------------------------------------------------------------------------------------------------------------
def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def spawn_ranks(n: int, argv) -> int:
    """`python bench.py --gpus N` without a launcher: re-execute as N ranks of one node, one GPU each."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.abspath(__file__), *argv]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def load_pmc_traffic(args):
    """HBM bytes per decode step from the committed rocprofv3 PMC passes of the same command line
    (tools/profile_round.sh -> profiles/*_pmc_step_traffic.json); None when no summary matches this configuration."""
    from thewhisper_amd.build import source_digest

    best = None
    sha = source_digest()
    for p in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_step_traffic.json"))):
        try:
            d = json.load(open(p))
        except Exception:  # noqa: BLE001
            continue
        c = d.get("config", {})
        if d.get("kernel_source_sha256") != sha:
            continue      # counters of OTHER kernels (an older round's build): not this run's traffic
        if (c.get("model"), c.get("streams"), c.get("chunk_s"), c.get("dtype"), c.get("new_tokens")) == \
                (args.model, args.streams, args.chunk_s, args.dtype, args.new_tokens):
            best = (d, os.path.basename(p))
    return best


class TimedRun:
    """One timed configuration: contexts (+ the encoder-overlap pipeline) of one dtype, W warm-up steps, K timed steps bracketed by
    barriers, per-stage HIP-event times.  A STEP = every stream of this GPU's share once through the whole hot path: `passes`
    engine passes of up to `B` streams each (weak scaling: one pass of --streams; strong scaling: ceil(share / 64) passes)."""

    def __init__(self, args, dims, dtype, rep, local, dev, stub, B, passes):
        self.args, self.dims, self.dtype, self.rep, self.B, self.passes = args, dims, dtype, rep, B, passes
        T = self.T = 50 * args.chunk_s
        heads = self.heads = alignment_heads(dims)
        if stub:
            mod, fn = stub.split(":")
            make_engine = getattr(importlib.import_module(mod), fn)
        else:
            from thewhisper_amd.engine import WhisperEngine

            def make_engine():
                return WhisperEngine(dims, T, max_batch=B, dtype=dtype, alignment_heads=heads, device=local, use_graph=not args.no_graph)

        self.eng = eng = make_engine()
        self.engines = [eng]
        sd = None if stub else random_state_dict(dims, dev, seed=0)
        if sd is not None:
            eng.load_state_dict(sd)
        self.overlap, self.overlap_note = None, "off"
        if args.encoder_cus > 0 and not stub:
            try:
                from thewhisper_amd.overlap import EncoderOverlap

                eng2 = make_engine()
                self.engines.append(eng2)
                eng2.load_state_dict(sd)
                self.overlap = EncoderOverlap([eng, eng2], encoder_cus=args.encoder_cus, decoder_cus=args.decoder_cus or None)
                self.overlap_note = (f"encoder stage of pass k+1 on {args.encoder_cus} CUs (second context) under the decode loop of "
                                     f"pass k on the other CUs")
            except Exception as e:  # noqa: BLE001 - the overlap is a schedule, not a requirement: run the stages back to back
                self.overlap = None
                self.overlap_note = f"off (could not set up CU-masked streams: {e!r})"
        del sd
        if not stub:
            torch.cuda.empty_cache()
        self.dev = dev
        self.batches = []      # set_share(): the streams of this GPU, `passes` batches of <= B
        self.prompt = np.tile(np.array([[50258, 50259, 50360]], dtype=np.int32), (B, 1))
        self.n_prompt = 3

    def set_share(self, share):
        """Cut this GPU's `share` streams into the batches of a step."""
        args, B = self.args, self.B
        g = torch.Generator(device=self.dev)
        g.manual_seed(1000 + self.rep.rank)
        self.share = share
        self.batches = []
        left = share
        while left > 0:
            nb = min(B, left)
            self.batches.append((torch.randn((nb, args.chunk_s * 16000), device=self.dev, generator=g, dtype=torch.float32) * 0.1).clamp_(-1, 1))
            left -= nb

    # -- the two stages of a pass (asynchronous launches; the greedy call blocks) -------------------------------------------
    @staticmethod
    def encode_stage(e, pc):      # log-mel + encoder + cross-K/V
        mel = e.logmel(pc)
        e.encode(mel)
        e.cross_kv(pc.shape[0])
        return mel                # kept alive until the batch is decoded: the launches above are still reading it

    def decode_stage(self, e, pc, _enc):
        args, nb, host = self.args, pc.shape[0], self.host
        t_a = time.perf_counter()
        out = e.generate_greedy(self.prompt[:nb], max_new_tokens=args.new_tokens, min_new_tokens=args.new_tokens,
                                timestamps=True, want_alignment=True)
        t_b = time.perf_counter()
        L = out["length"]
        e.token_timestamps(nb, self.n_prompt, L, [2 * self.T] * nb)
        t_c = time.perf_counter()
        tm = e.last_timings()
        t_d = time.perf_counter()
        host["greedy_call_ms"] += (t_b - t_a) * 1e3
        host["timestamps_call_ms"] += (t_c - t_b) * 1e3
        host["timings_call_ms"] += (t_d - t_c) * 1e3
        host["calls"] += 1
        return (L - self.n_prompt) * nb, tm

    def run_steps(self, n):
        """n steps = n x `passes` engine passes; with the overlap the encoder stage of pass i+1 runs on its own CUs while pass i
        decodes (same kernels, same results, different schedule)."""
        seq = self.batches * n
        if self.overlap is not None:
            return self.overlap.run(seq, self.encode_stage, self.decode_stage)
        res = []
        for pc in seq:
            self.encode_stage(self.eng, pc)
            res.append(self.decode_stage(self.eng, pc, None))
        return res

    def timed(self):
        args, rep = self.args, self.rep
        self.host = {"greedy_call_ms": 0.0, "timestamps_call_ms": 0.0, "timings_call_ms": 0.0, "calls": 0}
        if args.warmup > 0:  # with the overlap at least two passes, so that both contexts capture their step graph untimed
            self.run_steps(max(args.warmup, 2 if (self.overlap is not None and len(self.batches) == 1) else args.warmup))
        rep.barrier()                          # dist.barrier() + torch.cuda.synchronize()
        self.host.update(greedy_call_ms=0.0, timestamps_call_ms=0.0, timings_call_ms=0.0, calls=0)
        stage = {"logmel_ms": 0.0, "encode_ms": 0.0, "cross_kv_ms": 0.0, "greedy_ms": 0.0, "token_timestamps_ms": 0.0}
        dec_steps = new_tok = 0
        t0 = time.perf_counter()
        for ntok, tm in self.run_steps(args.steps):
            new_tok += ntok
            for k in stage:
                stage[k] += tm[k]
            dec_steps += tm["decode_steps"]
        rep.barrier()
        dt_local = time.perf_counter() - t0
        return {"dt_local": dt_local, "new_tok": new_tok, "stage": stage, "dec_steps": dec_steps, "host": dict(self.host)}

    def roofline(self, r, dt):
        """Decode step: SURVEY 8d's algorithmic bytes over the HIP-event time of the loop (per engine pass of B streams)."""
        args, dims, B, T = self.args, self.dims, self.B, self.T
        esz = 4 if self.dtype == "f32" else 2
        wsz = 1.0 + 1.0 / 32 if self.dtype.startswith("fp8") else None
        n_pass = max(1, args.steps * len(self.batches))
        steps_per_call = r["dec_steps"] // n_pass
        # (strong scaling: the passes of a step may differ in size; the byte count uses each pass's own stream count)
        alg_total = 0
        for pc in self.batches:
            a, W = algorithmic_decode_bytes(dims, pc.shape[0], T, self.n_prompt, steps_per_call, esz, wsz)
            alg_total += a
        alg_bytes, W = int(alg_total / len(self.batches)), int(W)
        greedy_ms = r["stage"]["greedy_ms"] / n_pass
        achieved = alg_bytes / (greedy_ms * 1e-3) / 1e9 if greedy_ms > 0 else 0.0
        return {"alg_bytes": alg_bytes, "W": W, "greedy_ms": greedy_ms, "achieved": achieved, "steps_per_call": steps_per_call,
                "avg_step_ms": greedy_ms / max(1, steps_per_call), "esz": esz, "wsz": wsz}

    def close(self):
        if self.overlap is not None:
            self.overlap.close()
        for e in self.engines:
            e.close()
        self.engines, self.batches = [], []


def full_depth_parity(dtype):
    """`streams_with_identical_ids` and friends of the headline-shaped parity case (16 clips x 163 positions at full depth, fp32
    reference goldens) from the newest committed log of the GPU suite: what the id-identity claim of a dtype rests on."""
    from thewhisper_amd.build import source_digest

    logs = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_gpu_tests_full_depth.log")))
    for path in reversed(logs):
        for line in open(path, errors="replace"):
            if line.startswith(f"FULLDEPTH full_large-v3_c10_b16 {dtype}:"):
                k = line.find("kernel_source_sha256=")
                sha = line[k + 21 :].split(",")[0].strip() if k >= 0 else None
                if sha != source_digest():
                    # figures of OTHER kernels are not this build's parity: never reported (tests/test_bench_cpu.py fails the CPU suite
                    # while the committed log is stale, so this branch is not reached by a tree whose tests are green)
                    msg = (f"profiles/{os.path.basename(path)} was taken with kernel sources {str(sha)[:12]}..., this build is "
                           f"{source_digest()[:12]}...: re-run tests/test_gpu_full_depth.py on the MI355X and commit its log")
                    print(f"[bench] STALE parity_full_depth ({dtype}): {msg}", file=sys.stderr, flush=True)
                    return {"stale": True, "error": msg}
                out = {"source": f"profiles/{os.path.basename(path)} (tests/test_gpu_full_depth.py, case full_large-v3_c10_b16: 16 clips, 160 free-running "
                                 f"greedy tokens each, HF fp32 reference)", "clips": 16}
                for key, name in (("streams_with_identical_ids", "ids_identical_clips"), ("greedy_path_logits_rel_l2", "logits_rel_l2"),
                                  ("token_ts_within_1_frame_frac", "token_timestamps_within_1_frame_frac"), ("enc_rel_l2", "encoder_rel_l2")):
                    k = line.find(key + "=")
                    if k >= 0:
                        v = line[k + len(key) + 1 :].split(",")[0].strip()
                        try:
                            out[name] = int(v) if name == "ids_identical_clips" else round(float(v), 6)
                        except ValueError:
                            pass
                return out
    return None


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--model", default="large-v3")
    ap.add_argument("--chunk-s", type=int, default=10)
    ap.add_argument("--streams", type=int, default=16, help="concurrent streams per GPU (1..64; 16 = the per-GPU share of BASELINE configs[3])")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak (default): --streams per GPU whatever N (configs[3]: 16 x N streams).  strong: --total-streams over ALL GPUs "
                         "(SURVEY.md section 8d row 4, 'fixed-128'): every GPU takes total / N of them, in passes of at most 64")
    ap.add_argument("--total-streams", type=int, default=128, help="with --scaling strong: streams of the whole job (BASELINE configs[3]: 128)")
    ap.add_argument("--new-tokens", type=int, default=128)
    ap.add_argument("--dtype", default="f16", choices=["bf16", "f16", "f32", "fp8", "fp8a8", "fp8a16"],
                    help="context element type.  f16 (default since round 6): the reference's streaming default "
                         "(R:thestage_speechkit/streaming/streaming_pipeline.py:369-370, torch.float16) and the 16-bit type whose greedy ids equal "
                         "the fp32 reference's on every full-depth clip (parity_full_depth); bf16 is reported beside it (value_bf16).  "
                         "fp8 = bf16 activations/encoder + MXFP8 decoder projection weights (BASELINE config 5)")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-tokens", type=int, default=32, help="forced tokens per CPU-baseline call (BASELINE.md section 3: >= 3 calls x 32)")
    ap.add_argument("--cpu-calls", type=int, default=3)
    ap.add_argument("--encoder-cus", type=int, default=96,
                    help="overlap the encoder of batch k+1 (confined to this many CUs, second context) with the decode loop of "
                         "batch k (thewhisper_amd/overlap.py); 0 = one context, strictly sequential stages.  96: the decode "
                         "launches have 160 or 320 workgroups, which 256 - 96 = 160 CUs take in exactly one or two rounds "
                         "(same box: 10 436 vs 10 340 tok/s with 64)")
    ap.add_argument("--decoder-cus", type=int, default=0, help="with --encoder-cus: compute units of the decode loop (0 = all the others)")
    ap.add_argument("--latency-iters", type=int, default=100, help="single-stream chunk calls timed for the p50 (after 10 warm-ups; SURVEY.md section 8d)")
    ap.add_argument("--no-pipeline-leg", action="store_true", help="skip the measurement through ASRPipeline / BatchingHub")
    ap.add_argument("--no-secondary", action="store_true", help="skip the compact legs for BASELINE configs 2 (turbo, 30 s, 1 stream) and 5 (fp8, 15 s) and the float16 block")
    ap.add_argument("--hub-rounds", type=int, default=12, help="requests per session in the hub measurement")
    ap.add_argument("--no-hub-two-cohorts", dest="hub_two_cohorts", action="store_false",
                    help="skip the hub measurement with twice as many sessions as rows per pass (hub_2x_*)")
    ap.add_argument("--hub-prefetch-cus", type=int, default=96,
                    help="compute units of the side stream that encodes arrivals while a hub pass decodes (0 = off; serving.py)")
    ap.add_argument("--no-streams-sweep", dest="streams_sweep", action="store_false",
                    help="skip the compact 32 / 64 / 2 x 64 streams-per-GPU legs (the single-GPU points of the strong-scaling curve)")
    ap.add_argument("--hub-short-tokens", type=int, default=24,
                    help="max_new_tokens of the second hub measurement (the <= 2 passes per buffer regime; 0 = skip)")
    args = ap.parse_args(argv)

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(spawn_ranks(args.gpus, argv))

    # TW_BENCH_ENGINE="module:callable": control-flow tests of the N > 1 path on a box without GPUs (tests/test_bench_cpu.py)
    stub = os.environ.get("TW_BENCH_ENGINE")
    if not stub and not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the hot path has no CPU fallback")
    from thewhisper_amd.dist import Replicas, shard_streams

    local = int(os.environ.get("TW_BENCH_DEVICE", os.environ.get("LOCAL_RANK", "0")))  # TW_BENCH_DEVICE: plumbing tests only
    dev = torch.device("cpu") if stub else torch.device("cuda", local)
    if not stub:
        torch.cuda.set_device(local)
    rep = Replicas(device=dev)  # nccl (= RCCL) when WORLD_SIZE > 1; barrier / max / sum only
    rank, world = rep.rank, rep.world
    if world != args.gpus:
        raise SystemExit(f"bench.py --gpus {args.gpus} but the process group has {world} rank(s): launch with "
                         f"`python bench.py --gpus {args.gpus}` or torch.distributed.run --nproc-per-node {args.gpus}")

    dims = DIMS[args.model]
    T = 50 * args.chunk_s
    if args.scaling == "strong":
        # streams sharded like sessions: stream i lives on rank i % world (dist.shard_streams); a GPU takes its share in passes of <= 64
        share = len(shard_streams(args.total_streams, rank, world))
        if share < 1:
            raise SystemExit(f"--total-streams {args.total_streams} leaves rank {rank} of {world} without a stream")
        passes = (share + 63) // 64
        B = (share + passes - 1) // passes           # equal passes (128 on one GPU: 2 x 64; 32 per GPU: 1 x 32)
    else:
        share, passes, B = args.streams, 1, args.streams
    run = TimedRun(args, dims, args.dtype, rep, local, dev, stub, B, passes)
    run.set_share(share)
    eng, heads, prompt, n_prompt = run.eng, run.heads, run.prompt, run.n_prompt
    r = run.timed()
    dt_local = r["dt_local"]
    dt = rep.max_float(dt_local)           # max over ranks
    per_rank = rep.gather_floats(r["new_tok"] / dt_local)   # tokens/s of every rank on its own clock
    new_tok = rep.sum_int(r["new_tok"])    # whole-job token count
    stage, host = r["stage"], r["host"]

    def step(nb):
        pc = run.batches[0][:nb]
        mel = eng.logmel(pc)
        eng.encode(mel)
        eng.cross_kv(nb)
        out = eng.generate_greedy(prompt[:nb], max_new_tokens=args.new_tokens, min_new_tokens=args.new_tokens,
                                  timestamps=True, want_alignment=True)
        L = out["length"]
        eng.token_timestamps(nb, n_prompt, L, [2 * T] * nb)
        return L - n_prompt

    # single-stream chunk latency (config 3 shape) on the raw engine: same context, B = 1, PCM already in HBM
    lat = []
    if rank == 0 and args.latency_iters > 0 and not stub:
        for _ in range(min(10, args.latency_iters)):   # warm-ups
            step(1)
        tick = {"logmel_ms": 0.0, "encode_ms": 0.0, "cross_kv_ms": 0.0, "greedy_ms": 0.0, "token_timestamps_ms": 0.0}
        for _ in range(args.latency_iters):
            torch.cuda.synchronize()
            a = time.perf_counter()
            step(1)
            torch.cuda.synchronize()
            lat.append((time.perf_counter() - a) * 1e3)
            tm = eng.last_timings()
            for k in tick:
                tick[k] += tm[k] / args.latency_iters

    if rank == 0:
        rf = run.roofline(r, dt)
        alg_bytes, W, greedy_ms, achieved, steps_per_call, avg_step_ms = (rf[k] for k in ("alg_bytes", "W", "greedy_ms", "achieved", "steps_per_call", "avg_step_ms"))
        esz, wsz = rf["esz"], rf["wsz"]
        n_pass = args.steps * passes
        args_for_pmc = argparse.Namespace(**{**vars(args), "streams": B})
        pmc = load_pmc_traffic(args_for_pmc)
        if args.scaling == "strong":
            workload = (f"whisper-{args.model}, {args.chunk_s} s chunks, {args.total_streams} concurrent streams over ALL GPUs (configs[3], fixed total: "
                        f"{share} on this GPU in {passes} pass(es) of {B}), {args.new_tokens} new tokens/stream forced (min=max), timestamp grammar + "
                        f"word-timestamp DTW on")
        else:
            workload = (f"whisper-{args.model}, {args.chunk_s} s chunks, {B} concurrent streams per GPU (configs[3] per-GPU share), "
                        f"{args.new_tokens} new tokens/stream forced (min=max), timestamp grammar + word-timestamp DTW on")
        result = {
            "metric": "transcription tokens/sec (node), whisper-large-v3 10s chunks" if args.model == "large-v3" and args.chunk_s == 10
            else f"transcription tokens/sec (node), whisper-{args.model} {args.chunk_s}s chunks",
            "value": round(new_tok / dt, 2),
            "unit": "tok/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3),
            "higher_is_better": True,
            "scaling": args.scaling,
            "vs_baseline": None,
            "dtype": (f"bf16 activations, MXFP8 (e4m3 + block scales) decoder weights [{args.dtype}]" if args.dtype.startswith("fp8") else args.dtype),
            "data": "synthetic 16 kHz gaussian audio (sigma 0.1), random-init weights of the named architecture",
            "config": {
                "workload": workload,
                "streams_per_gpu": share, "streams_per_pass": B, "passes_per_step": passes, "chunk_seconds": args.chunk_s, "new_tokens": args.new_tokens,
                "total_streams": (args.total_streams if args.scaling == "strong" else share * world),
                "parallelism": f"replicas x{world} (streams sharded, no collective on the data path)",
                "decode_step_graph": not args.no_graph,
                "encoder_overlap": run.overlap_note,
            },
            "per_rank_tok_per_s": [round(x, 1) for x in per_rank],
            "stage_ms_per_step": {k: round(v / args.steps, 3) for k, v in stage.items()},
            "decode_tok_per_s": round(B * args.new_tokens / (greedy_ms * 1e-3), 1) if greedy_ms > 0 else None,
            # wall time of the host calls per pass, beside the HIP-event time of the loop inside tw_generate_greedy (greedy_ms):
            # what ms_per_step holds besides the decode loop (call set-up / tear-down, DTW call, pipeline fill of the first batch)
            "host_call_ms_per_step": {k: round(v / max(1, host["calls"]) * passes, 3) for k, v in host.items() if k != "calls"},
            "roofline": {
                "kernel": "decode step (the captured graph replays two at a time; weight-streaming projections + single-query attention "
                          "over the K/V caches + sampler); averaged over all steps of a greedy call",
                "bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4),
                "frac_of_copy_ceiling": round(achieved / HBM_COPY_CEILING_GBS, 4), "copy_ceiling": HBM_COPY_CEILING_GBS,
                "traffic": (round(pmc[0]["hbm_bytes_per_step"]) if pmc else None),
                "traffic_source": (f"profiles/{pmc[1]}: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command line with THESE kernels "
                                   f"(kernel_source_sha256 matches the running build), FETCH_SIZE x2 (gfx950 correction, MI355X_MICROARCH.md)"
                                   if pmc else "no PMC summary under profiles/ for this workload (model, streams, chunk length, dtype, new tokens) taken with this build's kernel sources (tools/profile_round.sh)"),
                "algorithmic_bytes_per_step": int(alg_bytes / max(1, steps_per_call)),
                "algorithmic_bytes_per_call": alg_bytes, "weight_bytes_per_step": W, "decode_steps_per_call": steps_per_call,
                # what the launches really stream: the "cross query ahead" fusion reads 2 d^2 more weight elements per layer (DESIGN.md
                # section 4) than SURVEY's 14 d^2; `achieved` / `frac` stay on SURVEY's algorithmic bytes, `traffic` is the counter
                "weight_bytes_streamed_per_step": int(W + (wsz if wsz is not None else esz) * dims["dec_layers"] * 2 * dims["d_model"] ** 2),
                "avg_step_ms": round(avg_step_ms, 4),
                "streams_per_launch": B,
            },
        }
        if lat:
            lat.sort()
            result["p50_chunk_latency_ms"] = round(lat[len(lat) // 2], 2)
            result["p90_chunk_latency_ms"] = round(lat[min(len(lat) - 1, (len(lat) * 9) // 10)], 2)
            # SURVEY.md section 8f-3 (incremental streaming = reuse work between the 0.5 s ticks of one stream's rolling buffer): what a
            # tick costs, stage by stage.  Re-encoding the whole buffer - what an approximate encoder reuse could save, at the price
            # of different results - is the encode + cross-K/V share; the rest is the autoregressive loop over up to 128 tokens.
            enc_share = (tick["logmel_ms"] + tick["encode_ms"] + tick["cross_kv_ms"]) / max(1e-9, sum(tick.values()))
            result["streaming_tick_breakdown"] = {**{k: round(v, 3) for k, v in tick.items()},
                                                  "encoder_side_share": round(enc_share, 4),
                                                  "note": "one stream, one tick = one backend call on the rolling buffer (engine level); "
                                                          "8f-3 closed on this number: encoder reuse could remove at most encoder_side_share "
                                                          "of a tick while changing results (non-causal encoder), DESIGN.md section 7"}
            result["chunk_latency_note"] = (f"raw engine, 1 stream, {args.chunk_s} s chunk resident in HBM, {args.new_tokens} tokens + DTW, "
                                            f"{len(lat)} calls; the contract's latency (one backend call, host buffer in, words out) is "
                                            f"p50_chunk_latency_contract_ms")
        if world == 1 and not stub and not args.no_pipeline_leg and args.scaling == "weak":
            try:
                pargs = argparse.Namespace(**{**vars(args), "streams": B})
                result["pipeline"] = pipeline_leg(eng, dims, pargs, local, heads, args.latency_iters, args.hub_rounds)
            except Exception as e:  # noqa: BLE001
                result["pipeline"] = {"error": repr(e)}
        pl = result.get("pipeline") or {}
        # ---- the CONTRACT's definitions first (R:thestage_speechkit/streaming/streaming_pipeline.py:388-435), the
        # engine-level readings beside them
        result["value_contract"] = pl.get("hub_tok_per_s")
        result["p50_chunk_latency_contract_ms"] = pl.get("backend_transcribe_p50_ms")
        result["p90_chunk_latency_contract_ms"] = pl.get("backend_transcribe_p90_ms")
        result["hub_request_p50_ms"] = pl.get("hub_request_p50_ms")
        result["hub_request_p90_ms"] = pl.get("hub_request_p90_ms")
        result["config3"] = pl.get("config3")
        result["value_definition"] = ("value = tokens/s of the hot path driven through the C ABI with the PCM resident in HBM (what a host written "
                                      "against include/thewhisper.h gets; bench.py's contract: inputs resident when the timed region starts). "
                                      "value_contract (= value_api) = SURVEY.md section 8d's definition: generated tokens over wall-clock from "
                                      "transcribe() entry to return, host buffers in, word dictionaries out, 16 free-running sessions through "
                                      "AMDWhisperBackend / BatchingHub (pipeline.hub_tok_per_s).  p50_chunk_latency_contract_ms = median wall time of one "
                                      "AMDWhisperBackend.transcribe call on a 10 s host buffer (8d: 'one backend call'), p50_chunk_latency_ms = the raw "
                                      "engine on resident PCM.  config3 = the reference scheduler's own 60 s call trace replayed through the backend")
        result["value_api"] = pl.get("hub_tok_per_s")
        # the same API with two cohorts of sessions taking turns on the same passes (32 sessions for 16 rows): the encoder stage of the
        # cohort that sits a pass out runs on the side stream under that pass's decode loop
        result["value_api_two_cohorts"] = pl.get("hub_2x_tok_per_s")
        # value_api over the window in which every session still has requests to send (no drain passes at the end of the finite run)
        result["value_api_steady_state"] = pl.get("hub_steady_tok_per_s")
        result["value_api_two_cohorts_steady_state"] = pl.get("hub_2x_steady_tok_per_s")
        if world == 1 and not stub and not args.no_secondary:
            legs = []
            for label, model, chunk_s, nb, dt_, k in (("configs[1]: large-v3-turbo, 30 s chunk, batch 1, bf16", "large-v3-turbo", 30, 1, "bf16", 5),
                                                        ("configs[4]: large-v3, MXFP8 decoder weights + fp8 cross-K/V, 15 s chunks, word timestamps", "large-v3", 15, min(B, 16), "fp8", 3),
                                                        ("configs[4] shape in bf16 (for the fp8 / bf16 ratio)", "large-v3", 15, min(B, 16), "bf16", 3)):
                try:
                    legs.append(secondary_leg(label, model, chunk_s, nb, dt_, k, args.new_tokens, local, use_graph=not args.no_graph))
                except Exception as e:  # noqa: BLE001
                    legs.append({"config": label, "error": repr(e)})
            result["other_configs"] = legs
            try:   # the reference's own headline unit (RTFx of the turbo engines), its definitions, this backend's pipeline
                sys.path.insert(0, os.path.join(ROOT, "benchmark"))
                import run_rtfx

                rr = run_rtfx.measure("large-v3-turbo", minutes=4.0, batch_sizes=(1, 32), device_index=local)
                result["rtfx"] = {"definition": "audio seconds / wall seconds of ASRPipeline(audio, batch_size=bs) on one long clip "
                                                "(R:benchmark/eval_utils.py:149-154); TTFT = inference start -> first token",
                                  "model": rr["model"], "audio_s": rr["audio_s"], "forced_new_tokens_per_30s_window": rr["forced_new_tokens_per_window"],
                                  "runs": rr["runs"]}
            except Exception as e:  # noqa: BLE001
                result["rtfx"] = {"error": repr(e)}
        # ---- the OTHER 16-bit context as a first-class block.  Headline = float16 (round 6): the reference's streaming default dtype
        # (R:...streaming_pipeline.py:369-370) and the 16-bit context whose greedy ids are identical to the fp32 reference on every clip of
        # the full-depth suite; bfloat16 (the nvidia pipeline's alternative, R:thestage_speechkit/nvidia/asr_pipeline.py:47-60) runs the SAME
        # timed schedule (overlap pipeline, K steps) with its own roofline; the parity figures of both from the committed GPU-suite log
        result["parity_full_depth"] = {args.dtype: full_depth_parity(args.dtype)} if args.dtype in ("bf16", "f16", "f32") else {}
        if args.dtype == "f16":
            result["dtype_note"] = ("float16 is the headline dtype since round 6: it is the reference's streaming default (torch.float16, "
                                    "streaming_pipeline.py:369-370), costs what bf16 costs (value_bf16 beside it) and carries the north star's "
                                    "'greedy token IDs identical' claim at full depth (parity_full_depth.f16.ids_identical_clips); HF's own bf16 "
                                    "arithmetic flips sub-margin decisions too (tests/test_golden_ctrl.py)")
        other = {"f16": "bf16", "bf16": "f16"}.get(args.dtype)
        if world == 1 and not stub and not args.no_secondary and other is not None:
            run.close()
            run = None
            torch.cuda.empty_cache()
            try:
                run16 = TimedRun(args, dims, other, rep, local, dev, stub, B, passes)
                run16.set_share(share)
                r16 = run16.timed()
                rf16 = run16.roofline(r16, r16["dt_local"])
                result[f"value_{other}"] = round(r16["new_tok"] / r16["dt_local"], 2)
                result[f"ms_per_step_{other}"] = round(r16["dt_local"] / args.steps * 1e3, 3)
                result[f"roofline_{other}"] = {"bound": "hbm", "achieved": round(rf16["achieved"], 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                               "frac": round(rf16["achieved"] / HBM_PEAK_GBS, 4), "avg_step_ms": round(rf16["avg_step_ms"], 4),
                                               "algorithmic_bytes_per_step": int(rf16["alg_bytes"] / max(1, rf16["steps_per_call"])), "traffic": None}
                result[f"stage_ms_per_step_{other}"] = {k: round(v / args.steps, 3) for k, v in r16["stage"].items()}
                result["parity_full_depth"][other] = full_depth_parity(other)
                result[f"{other}_note"] = (f"the headline workload in a {other} context (the same kernels instantiated for the other 16-bit type: "
                                           "v_mfma_f32_16x16x32_bf16 / _f16), same schedule, same step count; parity_full_depth.*.ids_identical_clips = "
                                           "clips (of 16) whose 160 free-running greedy ids equal the HF fp32 reference's to the end")
                run16.close()
            except Exception as e:  # noqa: BLE001
                result[f"value_{other}"] = None
                result[f"{other}_note"] = f"failed: {e!r}"
        # ---- streams per GPU: what ONE MI355X delivers with 32 and 64 streams per weight pass, and with BASELINE configs[3]'s whole 128
        # streams as two passes of 64 = the N = 1 point of the strong-scaling curve.  Replicas share nothing, so
        # the other points of that curve are these per-GPU rates times N (128 / N streams per GPU): N = 2 -> 2 x the 64-stream rate,
        # N = 4 -> 4 x the 32-stream rate, N = 8 -> 8 x the headline; a node run with `--scaling strong` measures them directly.
        if world == 1 and not stub and not args.no_secondary and args.streams_sweep and args.scaling == "weak" and args.streams == 16:
            if run is not None:
                run.close()
                run = None
                torch.cuda.empty_cache()
            sweep = []
            sargs = argparse.Namespace(**{**vars(args), "steps": 3, "warmup": 1})
            for label, sh in (("32 streams per GPU", 32), ("64 streams per GPU", 64), ("128 streams on this GPU, two passes of 64 (strong scaling, N = 1)", 128)):
                try:
                    ps = (sh + 63) // 64
                    r_s = TimedRun(sargs, dims, args.dtype, rep, local, dev, stub, (sh + ps - 1) // ps, ps)
                    r_s.set_share(sh)
                    t_s = r_s.timed()
                    f_s = r_s.roofline(t_s, t_s["dt_local"])
                    sweep.append({"workload": label, "streams": sh, "passes_per_step": ps, "tok_per_s": round(t_s["new_tok"] / t_s["dt_local"], 1),
                                  "ms_per_step": round(t_s["dt_local"] / sargs.steps * 1e3, 2), "decode_step_ms": round(f_s["avg_step_ms"], 4),
                                  "roofline_frac": round(f_s["achieved"] / HBM_PEAK_GBS, 4), "steps": sargs.steps})
                    r_s.close()
                    torch.cuda.empty_cache()
                except Exception as e:  # noqa: BLE001
                    sweep.append({"workload": label, "error": repr(e)})
            result["streams_sweep"] = sweep
            ok = {x["streams"]: x["tok_per_s"] for x in sweep if "tok_per_s" in x}
            if {32, 64, 128} <= set(ok):
                result["strong_scaling_128_streams"] = {
                    "n_gpus_1_measured": ok[128],
                    "projected_from_per_gpu_rates": {"2": round(2 * ok[64], 1), "4": round(4 * ok[32], 1), "8": round(8 * result["value"], 1)},
                    "note": "128 concurrent streams (BASELINE configs[3]) on N GPUs; N = 1 is measured here (two passes of 64), N = 2 / 4 / 8 are N x the "
                            "measured single-GPU rate at 64 / 32 / 16 streams per GPU (replicas, no data-path collective: DESIGN.md section 5) - a "
                            "projection, NOT a measurement; `python bench.py --gpus N --scaling strong --total-streams 128` measures it on a node"}
        if world == 1 and not stub and not args.no_cpu_baseline:
            try:
                result["cpu_baseline"] = cpu_baseline(args.model, args.chunk_s, args.cpu_tokens, args.cpu_calls)
            except Exception as e:  # noqa: BLE001
                result["cpu_baseline"] = {"value": None, "unit": "tok/s", "cores": os.cpu_count(), "kind": "reference",
                                          "sample": f"failed: {e!r}"}
        print(json.dumps(result), flush=True)
    if run is not None:
        run.close()
    rep.close()


if __name__ == "__main__":
    main()
