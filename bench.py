#!/usr/bin/env python3
"""bench.py - BASELINE.json's headline metric on MI355X: transcription tokens/sec, whisper-large-v3, 10 s chunks.

One "step" = one pass of the whole hot path over one batch of synthetic 16 kHz audio that is already
resident in HBM:  log-mel -> encoder -> cross-K/V -> 128-token greedy decode (timestamp grammar on, as the
reference's streaming backend always runs it, R:thestage_speechkit/streaming/streaming_pipeline.py:395-410)
-> alignment DTW (word timestamps).  Per GPU the workload is `--streams` (default 16) concurrent 10 s
chunks = configs[3]'s per-GPU share (128 streams / 8 GPUs); N GPUs run N independent replicas on
disjoint streams (weak scaling, no data-path collective - SURVEY.md section 8e).

Launch: `python bench.py` (1 GPU) or
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...`
Rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import dataclasses
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

DIMS = {
    "large-v3": dict(d_model=1280, enc_layers=32, dec_layers=32, heads=20, ffn=5120, vocab=51866, n_mels=128,
                     max_source_positions=1500, max_target_positions=448),
    "large-v3-turbo": dict(d_model=1280, enc_layers=32, dec_layers=4, heads=20, ffn=5120, vocab=51866, n_mels=128,
                           max_source_positions=1500, max_target_positions=448),
    "tiny.en": dict(d_model=384, enc_layers=4, dec_layers=4, heads=6, ffn=1536, vocab=51864, n_mels=80,
                    max_source_positions=1500, max_target_positions=448),
}
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured float4 copy)


def random_state_dict(dims, device, seed=0):
    """Random-init weights of the named architecture, generated on the GPU in the HF state_dict layout."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    d, f, v = dims["d_model"], dims["ffn"], dims["vocab"]

    def uni(shape, amp):
        return (torch.rand(shape, device=device, generator=g, dtype=torch.float32) - 0.5) * (2 * amp)

    sd = {}

    def lin(name, o, i, bias=True):
        sd[name + ".weight"] = uni((o, i), 1.7 / i ** 0.5)
        if bias:
            sd[name + ".bias"] = uni((o,), 0.05)

    def ln(name):
        sd[name + ".weight"] = 1.0 + uni((d,), 0.1)
        sd[name + ".bias"] = uni((d,), 0.05)

    def attn(p):
        lin(p + ".k_proj", d, d, False)
        lin(p + ".v_proj", d, d)
        lin(p + ".q_proj", d, d)
        lin(p + ".out_proj", d, d)

    e = "model.encoder"
    sd[e + ".conv1.weight"] = uni((d, dims["n_mels"], 3), 1.7 / (3 * dims["n_mels"]) ** 0.5)
    sd[e + ".conv1.bias"] = uni((d,), 0.05)
    sd[e + ".conv2.weight"] = uni((d, d, 3), 1.7 / (3 * d) ** 0.5)
    sd[e + ".conv2.bias"] = uni((d,), 0.05)
    sd[e + ".embed_positions.weight"] = uni((dims["max_source_positions"], d), 0.5)
    for i in range(dims["enc_layers"]):
        p = f"{e}.layers.{i}"
        attn(p + ".self_attn"); ln(p + ".self_attn_layer_norm"); lin(p + ".fc1", f, d); lin(p + ".fc2", d, f); ln(p + ".final_layer_norm")
    ln(e + ".layer_norm")
    dd = "model.decoder"
    sd[dd + ".embed_tokens.weight"] = uni((v, d), 0.12)
    sd[dd + ".embed_positions.weight"] = uni((dims["max_target_positions"], d), 0.12)
    for i in range(dims["dec_layers"]):
        p = f"{dd}.layers.{i}"
        attn(p + ".self_attn"); ln(p + ".self_attn_layer_norm"); attn(p + ".encoder_attn"); ln(p + ".encoder_attn_layer_norm")
        lin(p + ".fc1", f, d); lin(p + ".fc2", d, f); ln(p + ".final_layer_norm")
    ln(dd + ".layer_norm")
    return sd


def alignment_heads(dims):
    n = min(10, max(2, dims["dec_layers"] * 2))
    out = []
    for j in range(n):
        layer = dims["dec_layers"] - 1 - (j % max(1, dims["dec_layers"] // 2))
        head = (3 * j + 1) % dims["heads"]
        if (layer, head) not in out:
            out.append((layer, head))
    return out


def algorithmic_decode_bytes(dims, B, T, n_prompt, steps, esz=2, wsz=None):
    """bytes/step = W + B*163840*(T + t) (large-v3 numbers generalised):
    W = esz*(Ld*14*d^2 + V*d); per stream and step the cross K/V (2*Ld*T*d*esz) and the self K/V read so far."""
    d, Ld, V = dims["d_model"], dims["dec_layers"], dims["vocab"]
    wsz = esz if wsz is None else wsz  # bytes per projection weight (MXFP8: 1 + 1/32 for the block scales)
    W = wsz * (Ld * (4 * d * d + 4 * d * d + 2 * d * dims["ffn"]) + V * d)
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


def _cpu_baseline_worker(model_name, chunk_s, new_tokens):
    """Child process: the reference's PyTorch-CPU arithmetic (HF transformers generate, fp32, all host cores)."""
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
    for name, p in model.named_parameters():   # timing does not depend on the values: cheap deterministic fill
        if p.dim() >= 2:
            p.fill_(0.5 / p.shape[-1])
        elif "layer_norm" in name and name.endswith("weight"):
            p.fill_(1.0)
        else:
            p.zero_()
    model.eval()
    hr.fill_generation_config(model.generation_config, dims)
    hr.patch_chunk_length(model, chunk_s)
    fe = hr.build_feature_extractor(dims, chunk_s)
    t_init = time.time() - t0
    pcm = wo.synth_audio(chunk_s * 16000, 0, "noise")
    fe(pcm[:16000], sampling_rate=16000, return_tensors="pt")   # untimed: first-call cost of torch.stft
    t0 = time.time()
    feats = fe(pcm, sampling_rate=16000, return_tensors="pt", return_attention_mask=True)
    model.generate(input_features=feats.input_features, attention_mask=feats.attention_mask, language="en",
                   return_timestamps=True, num_beams=1, do_sample=False, use_cache=True,
                   max_new_tokens=new_tokens, min_new_tokens=new_tokens, force_unique_generate_call=True)
    dt = time.time() - t0
    print("CPU_BASELINE " + json.dumps({"tok": new_tokens, "dt": dt, "init": t_init, "cores": cores}), flush=True)


def cpu_baseline(model_name, chunk_s, new_tokens, budget_s=150):
    """Reported baseline (not the target): the reference's CPU path on a BOUNDED sample - one `chunk_s` s chunk,
    `new_tokens` forced tokens - in a child process with a hard wall-clock budget.  oracle/ is imported only here."""
    import subprocess

    cores = _host_cores()
    code = f"import bench; bench._cpu_baseline_worker({model_name!r}, {chunk_s}, {new_tokens})"
    env = dict(os.environ, HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES="")
    what = (f"HF transformers WhisperForConditionalGeneration.generate on CPU fp32 (the arithmetic behind the reference's nvidia HF "
            f"branch, R:thestage_speechkit/nvidia/asr_pipeline.py:57-60), {model_name} dims, 1 stream x {chunk_s} s chunk: "
            f"log-mel + encoder + {new_tokens} forced tokens")
    try:
        r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=env, capture_output=True, text=True, timeout=budget_s)
        line = [l for l in r.stdout.splitlines() if l.startswith("CPU_BASELINE ")]
        if not line:
            raise RuntimeError((r.stderr or r.stdout)[-300:])
        d = json.loads(line[-1][len("CPU_BASELINE "):])
        return {"value": round(d["tok"] / d["dt"], 3), "unit": "tok/s", "cores": d["cores"], "kind": "reference",
                "sample": what + f", {d['dt']:.1f} s wall (+{d['init']:.1f} s model init)"}
    except subprocess.TimeoutExpired:
        return {"value": round(new_tokens / budget_s, 3), "unit": "tok/s", "cores": cores, "kind": "reference",
                "sample": what + f": did NOT finish within the {budget_s} s budget - value is an upper bound"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--model", default="large-v3")
    ap.add_argument("--chunk-s", type=int, default=10)
    ap.add_argument("--streams", type=int, default=16, help="concurrent streams per GPU (1..64; 16 = the per-GPU share of BASELINE configs[3])")
    ap.add_argument("--new-tokens", type=int, default=128)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32", "fp8"],
                    help="fp8 = bf16 activations/encoder + MXFP8 decoder projection weights (BASELINE config 5)")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-tokens", type=int, default=8)
    ap.add_argument("--encoder-cus", type=int, default=64,
                    help="overlap the encoder of batch k+1 (confined to this many CUs, second context) with the decode loop of "
                         "batch k (thewhisper_amd/overlap.py); 0 = one context, strictly sequential stages")
    ap.add_argument("--latency-iters", type=int, default=100, help="single-stream chunk calls timed for the p50 (after 10 warm-ups; SURVEY.md section 8d)")
    args = ap.parse_args()

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the hot path has no CPU fallback")
    from thewhisper_amd.dist import Replicas

    local = int(os.environ.get("TW_BENCH_DEVICE", os.environ.get("LOCAL_RANK", "0")))  # TW_BENCH_DEVICE: plumbing tests only
    torch.cuda.set_device(local)
    rep = Replicas(device=torch.device("cuda", local))  # nccl (= RCCL) when WORLD_SIZE > 1; barrier / max / sum only
    rank, world = rep.rank, rep.world

    from thewhisper_amd.engine import WhisperEngine

    dims = DIMS[args.model]
    T = 50 * args.chunk_s
    B = args.streams
    dev = torch.device("cuda", local)
    heads = alignment_heads(dims)
    eng = WhisperEngine(dims, T, max_batch=B, dtype=args.dtype, alignment_heads=heads, device=local,
                        use_graph=not args.no_graph)
    sd = random_state_dict(dims, dev, seed=0)
    eng.load_state_dict(sd)
    overlap = None
    overlap_note = "off"
    if args.encoder_cus > 0:
        try:
            from thewhisper_amd.overlap import EncoderOverlap
            eng2 = WhisperEngine(dims, T, max_batch=B, dtype=args.dtype, alignment_heads=heads, device=local,
                                 use_graph=not args.no_graph)
            eng2.load_state_dict(sd)
            overlap = EncoderOverlap([eng, eng2], encoder_cus=args.encoder_cus)
            overlap_note = (f"encoder stage of batch k+1 on {args.encoder_cus} CUs (second context) under the decode loop of "
                            f"batch k on the other CUs")
        except Exception as e:  # noqa: BLE001 - the overlap is a schedule, not a requirement: run the stages back to back
            overlap = None
            overlap_note = f"off (could not set up CU-masked streams: {e!r})"
    del sd
    torch.cuda.empty_cache()

    n_samples = args.chunk_s * 16000
    g = torch.Generator(device=dev)
    g.manual_seed(1000 + rank)
    pcm = (torch.randn((B, n_samples), device=dev, generator=g, dtype=torch.float32) * 0.1).clamp_(-1, 1)
    prompt = np.tile(np.array([[50258, 50259, 50360]], dtype=np.int32), (B, 1))
    n_prompt = prompt.shape[1]

    def step(nb=B, pc=pcm, pr=prompt):
        mel = eng.logmel(pc[:nb])
        eng.encode(mel)
        eng.cross_kv(nb)
        out = eng.generate_greedy(pr[:nb], max_new_tokens=args.new_tokens, min_new_tokens=args.new_tokens,
                                  timestamps=True, want_alignment=True)
        L = out["length"]
        eng.token_timestamps(nb, n_prompt, L, [2 * T] * nb)
        return L - n_prompt

    def barrier():
        rep.barrier()                      # dist.barrier() + torch.cuda.synchronize()

    stage = {"logmel_ms": 0.0, "encode_ms": 0.0, "cross_kv_ms": 0.0, "greedy_ms": 0.0, "token_timestamps_ms": 0.0}
    dec_steps = 0
    new_tok = 0

    def encode_stage(e, pc):      # log-mel + encoder + cross-K/V: asynchronous launches
        mel = e.logmel(pc)
        e.encode(mel)
        e.cross_kv(pc.shape[0])
        return mel                # kept alive until the batch is decoded: the launches above are still reading it

    def decode_stage(e, pc, _enc):
        nb = pc.shape[0]
        out = e.generate_greedy(prompt[:nb], max_new_tokens=args.new_tokens, min_new_tokens=args.new_tokens,
                                timestamps=True, want_alignment=True)
        L = out["length"]
        e.token_timestamps(nb, n_prompt, L, [2 * T] * nb)
        return (L - n_prompt) * nb, e.last_timings()

    def run_steps(n):
        """n passes of the hot path over one batch each; with the overlap the encoder stage of pass i+1 runs on its own CUs
        while pass i decodes (same kernels, same results, different schedule)."""
        if overlap is not None:
            return overlap.run([pcm] * n, encode_stage, decode_stage)
        res = []
        for _ in range(n):
            encode_stage(eng, pcm)
            res.append(decode_stage(eng, pcm, None))
        return res

    if args.warmup > 0:  # with the overlap at least two passes, so that both contexts capture their step graph untimed
        run_steps(max(args.warmup, 2) if overlap is not None else args.warmup)
    barrier()
    t0 = time.perf_counter()
    for ntok, tm in run_steps(args.steps):
        new_tok += ntok
        for k in stage:
            stage[k] += tm[k]
        dec_steps += tm["decode_steps"]
    barrier()
    dt = time.perf_counter() - t0
    dt = rep.max_float(dt)                 # max over ranks
    new_tok = rep.sum_int(new_tok)         # whole-job token count

    # single-stream chunk latency (config 3 shape): same engine, B = 1
    lat = []
    if rank == 0 and args.latency_iters > 0:
        for _ in range(min(10, args.latency_iters)):   # warm-ups
            step(1)
        for _ in range(args.latency_iters):
            torch.cuda.synchronize()
            a = time.perf_counter()
            step(1)
            torch.cuda.synchronize()
            lat.append((time.perf_counter() - a) * 1e3)

    if rank == 0:
        esz = 4 if args.dtype == "f32" else 2
        wsz = 1.0 + 1.0 / 32 if args.dtype == "fp8" else None
        steps_per_call = dec_steps // max(1, args.steps)
        alg_bytes, W = algorithmic_decode_bytes(dims, B, T, n_prompt, steps_per_call, esz, wsz)
        alg_bytes, W = int(alg_bytes), int(W)
        greedy_ms = stage["greedy_ms"] / args.steps
        achieved = alg_bytes / (greedy_ms * 1e-3) / 1e9 if greedy_ms > 0 else 0.0
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
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "bf16 activations, MXFP8 (e4m3 + block scales) decoder weights" if args.dtype == "fp8" else args.dtype,
            "data": "synthetic 16 kHz gaussian audio (sigma 0.1), random-init weights of the named architecture",
            "config": {
                "workload": f"whisper-{args.model}, {args.chunk_s} s chunks, {B} concurrent streams per GPU (configs[3] per-GPU share), "
                            f"{args.new_tokens} new tokens/stream forced (min=max), timestamp grammar + word-timestamp DTW on",
                "streams_per_gpu": B, "chunk_seconds": args.chunk_s, "new_tokens": args.new_tokens,
                "parallelism": f"replicas x{world} (streams sharded, no collective on the data path)",
                "decode_step_graph": not args.no_graph,
                "encoder_overlap": overlap_note,
            },
            "stage_ms_per_step": {k: round(v / args.steps, 3) for k, v in stage.items()},
            "decode_tok_per_s": round(B * args.new_tokens / (greedy_ms * 1e-3), 1) if greedy_ms > 0 else None,
            "roofline": {
                "kernel": "decode step (weight-streaming gemv + single-query attention over the KV caches), all steps of one call",
                "bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": None,
                "algorithmic_bytes_per_call": alg_bytes, "weight_bytes_per_step": W, "launches_per_call": steps_per_call,
                "avg_launch_ms": round(greedy_ms / max(1, steps_per_call), 4),
            },
        }
        if lat:
            lat.sort()
            result["p50_chunk_latency_ms"] = round(lat[len(lat) // 2], 2)
            result["p90_chunk_latency_ms"] = round(lat[min(len(lat) - 1, (len(lat) * 9) // 10)], 2)
            result["chunk_latency_note"] = f"1 stream, {args.chunk_s} s chunk, {args.new_tokens} tokens + DTW, {len(lat)} calls"
        if world == 1 and not args.no_cpu_baseline:
            try:
                result["cpu_baseline"] = cpu_baseline(args.model, args.chunk_s, args.cpu_tokens)
            except Exception as e:  # noqa: BLE001
                result["cpu_baseline"] = {"value": None, "unit": "tok/s", "cores": os.cpu_count(), "kind": "reference",
                                          "sample": f"failed: {e!r}"}
        print(json.dumps(result), flush=True)
    rep.close()


if __name__ == "__main__":
    main()
