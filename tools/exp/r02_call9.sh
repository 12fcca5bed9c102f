#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
rm -f $OUT/c9_*
for cfg in "TW_GEMM_NARROW=8" "TW_GEMM_NARROW=5" "TW_GEMM_NARROW=4"; do
  ( env $cfg timeout 600 python tools/bench_encoder.py --cases 500x16,750x16 2>&1 | grep "^T=" ) >> $OUT/c9_encoder_variants.log
done
( timeout 600 python tools/bench_encoder.py --cases 1500x1,500x1,500x4 2>&1 | grep "^T=" ) >> $OUT/c9_encoder_variants.log
python bench.py --chunk-s 15 --no-cpu-baseline --no-pipeline-leg --latency-iters 20 > $OUT/c9_bench_bf16_15s.json 2>/dev/null
python bench.py --dtype fp8 --chunk-s 15 --no-cpu-baseline --no-pipeline-leg --latency-iters 20 > $OUT/c9_bench_fp8_15s.json 2>/dev/null
cat $OUT/c9_encoder_variants.log
python - <<'PY'
import json
for f in ("c9_bench_bf16_15s.json","c9_bench_fp8_15s.json"):
    try:
        d=json.load(open("gpurun_out/"+f)); print(f, d["value"], d["ms_per_step"], d["stage_ms_per_step"], d["roofline"]["avg_step_ms"], d.get("p50_chunk_latency_ms"))
    except Exception as e: print(f, "ERR", e)
PY
