#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
rm -f $OUT/c10_*
( timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "mxfp8" 2>&1 | tail -30 ) > $OUT/c10_fp8_parity.log
( timeout 900 python -m pytest tests/test_gpu_full_depth.py -x -q -m gpu -s -k "fp8" 2>&1 | grep -E "FULLDEPTH|passed|failed|rror" ) > $OUT/c10_fp8_full.log
python bench.py --dtype fp8 --chunk-s 15 --no-cpu-baseline --no-pipeline-leg --latency-iters 20 > $OUT/c10_bench_fp8_15s.json 2>$OUT/c10_bench_fp8.err
python bench.py --dtype fp8 --no-cpu-baseline --no-pipeline-leg --latency-iters 20 > $OUT/c10_bench_fp8_10s.json 2>/dev/null
cat $OUT/c10_fp8_parity.log $OUT/c10_fp8_full.log | cut -c1-300
python - <<'PY'
import json
for f in ("c10_bench_fp8_15s.json","c10_bench_fp8_10s.json"):
    try:
        d=json.load(open("gpurun_out/"+f)); print(f, d["value"], d["ms_per_step"], d["stage_ms_per_step"], d["roofline"]["avg_step_ms"], d.get("p50_chunk_latency_ms"))
    except Exception as e: print(f, "ERR", e)
PY
tail -3 $OUT/c10_bench_fp8.err
