#!/bin/bash
# round 5, call 10: the whole GPU suite on the final kernels (FULLDEPTH lines -> profiles/r05_gpu_tests_full_depth.log), the round profile
# (tools/profile_round.sh), the streams sweep and the strong-scaling line
O=gpurun_out/r05_call10; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -q -s > $O/gpu_tests.log 2>&1; echo "pytest rc=$?" >> $O/gpu_tests.log
tail -4 $O/gpu_tests.log
bash tools/profile_round.sh r05 > $O/profile_round.log 2>&1
for B in 32 64; do timeout 600 python bench.py --streams $B --steps 5 --no-cpu-baseline --no-pipeline-leg --no-secondary --latency-iters 0 > gpurun_out/r05_bench_b$B.json 2>/dev/null; done
timeout 600 python bench.py --scaling strong --total-streams 128 --steps 5 --no-cpu-baseline --no-secondary --latency-iters 0 > gpurun_out/r05_bench_strong_128.json 2>/dev/null
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r05_bench_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f, d["value"], d["ms_per_step"], d["roofline"]["avg_step_ms"], d["roofline"]["frac"], d.get("value_contract"), d.get("value_f16"))
    except Exception as e:
        print(f, "ERR", e)
PY
