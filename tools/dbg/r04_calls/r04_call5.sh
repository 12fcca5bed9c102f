#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_reuse_prefix.py tests/test_gpu_serving.py tests/test_from_pretrained.py -m gpu -q -s 2>&1 | grep -E "REUSE|HUB PREFETCH|passed|failed|FAILED|Error" | tail -12
timeout 1200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --latency-iters 30 --hub-rounds 6 > $OUT/r04_c5_bench.json 2> $OUT/r04_c5_bench.err
python - <<PY
import json
d = json.load(open("$OUT/r04_c5_bench.json"))
p = d.get("pipeline") or {}
print("value", d["value"], "step", d["roofline"]["avg_step_ms"], "value_api", d.get("value_api"), "p50", d.get("p50_chunk_latency_ms"))
print({k: v for k, v in p.items() if ("reuse" in k or "scheduler" in k or "backend" in k) and "note" not in k})
PY
tail -5 $OUT/r04_c5_bench.err
