"""bench.py's N > 1 control flow without GPUs: `python bench.py --gpus 2` self-spawns two ranks (torch.distributed.run,
gloo), each drives a stub engine; rank 0 prints ONE JSON line with n_gpus = 2, per-rank rates and the whole-job sum."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra, env_extra=None):
    env = dict(os.environ, TW_BENCH_ENGINE="tests.stub_engine:make", TW_DIST_BACKEND="gloo", PYTHONPATH=ROOT,
               HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES="")
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", *extra],
                          cwd=ROOT, env=env, capture_output=True, text=True, timeout=280)


@pytest.mark.timeout(300)
def test_bench_self_spawns_ranks_and_prints_one_json_line():
    r = _run(["--gpus", "2"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["warmup"] == 1 and d["scaling"] == "weak"
    assert len(d["per_rank_tok_per_s"]) == 2 and all(x > 0 for x in d["per_rank_tok_per_s"])
    # whole-job value: 2 ranks x 2 steps x 16 streams x 128 tokens over the slowest rank's time
    assert abs(d["value"] * d["ms_per_step"] * 1e-3 * d["steps"] - 2 * 2 * 16 * 128) < 1.0
    assert d["roofline"]["decode_steps_per_call"] == 130 and "cpu_baseline" not in d and "pipeline" not in d
    # SURVEY 8d bytes: W = 2*(32*14*1280^2 + 51866*1280)
    assert d["roofline"]["weight_bytes_per_step"] == 2 * (32 * 14 * 1280 * 1280 + 51866 * 1280)


@pytest.mark.timeout(120)
def test_bench_refuses_a_rank_count_mismatch():
    r = _run(["--gpus", "2"], {"WORLD_SIZE": "1", "RANK": "0"})     # a launcher gave one rank, the command line says two
    assert r.returncode != 0 and "process group has 1 rank" in (r.stderr + r.stdout)


def test_algorithmic_bytes_follow_survey_8d():
    sys.path.insert(0, ROOT)
    import bench

    total, W = bench.algorithmic_decode_bytes(bench.DIMS["large-v3"], 16, 500, 3, 130)
    assert W == 1601044480 + 0 * 1 or abs(W - 1.601e9) < 1e6
    assert abs(total - 400.8e9) < 0.2e9      # the figure the round-1 review derived for the driver's call


@pytest.mark.timeout(300)
@pytest.mark.parametrize("gpus,share,passes,per_pass", [(2, 64, 1, 64), (1, 128, 2, 64)])
def test_bench_strong_scaling_mode_shards_a_fixed_total(gpus, share, passes, per_pass):
    """SURVEY.md section 8d row 4 ("fixed-128 for strong scaling"): --scaling strong --total-streams 128 gives every rank
    128 / N streams, taken in passes of at most 64; the whole-job value counts all 128 streams whatever N."""
    r = _run(["--gpus", str(gpus), "--scaling", "strong", "--total-streams", "128"])
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert d["scaling"] == "strong" and d["n_gpus"] == gpus
    c = d["config"]
    assert (c["total_streams"], c["streams_per_gpu"], c["passes_per_step"], c["streams_per_pass"]) == (128, share, passes, per_pass)
    assert abs(d["value"] * d["ms_per_step"] * 1e-3 * d["steps"] - d["steps"] * 128 * 128) < 1.0     # 128 streams x 128 tokens per step
    assert d["roofline"]["streams_per_launch"] == per_pass


def test_parity_block_reads_the_committed_full_depth_log():
    """bench.py's `parity_full_depth` block: the id-identity figures of a dtype come from the newest committed GPU-suite log."""
    sys.path.insert(0, ROOT)
    import bench

    f16, bf16 = bench.full_depth_parity("f16"), bench.full_depth_parity("bf16")
    # (round 6) the log names the kernel sources it was taken with; figures of other kernels are refused - THIS is where a stale log
    # fails loudly: re-run tests/test_gpu_full_depth.py on the MI355X after any change under thewhisper_amd/csrc and commit the log
    assert f16 and bf16 and not f16.get("stale") and not bf16.get("stale"), (f16, bf16)
    assert f16["clips"] == 16
    assert f16["ids_identical_clips"] == 16                       # the float16 context carries the id-identity claim
    assert 0 < bf16["ids_identical_clips"] <= 16 and bf16["logits_rel_l2"] > f16["logits_rel_l2"]
    assert "gpu_tests_full_depth.log" in f16["source"]
    assert bench.full_depth_parity("int3") is None


def test_config3_trace_is_where_the_bench_expects_it():
    tr = json.load(open(os.path.join(ROOT, "tests", "golden", "config3_trace.json")))
    assert tr["chunk_length_s"] == 10 and len(tr["calls"]) == 117 and all(set(c) == {"offset", "n", "t0"} for c in tr["calls"])
