#!/bin/bash
# round 4, call 14: where the generate call's wall time goes besides the device loop (TW_HOST_TIMING), with and without the overlap
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
for A in "" "--encoder-cus 0"; do
  echo "== $A"
  TW_HOST_TIMING=1 timeout 600 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-secondary --no-pipeline-leg --latency-iters 0 $A 2> $OUT/r04_c14_err.txt | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['stage_ms_per_step'], d['host_call_ms_per_step'])"
  grep TW_HOST_TIMING $OUT/r04_c14_err.txt | tail -4
done
for G in 2 4; do
  echo "== TW_GRAPH_STEPS_FORCED=$G"
  TW_GRAPH_STEPS_FORCED=$G TW_HOST_TIMING=1 timeout 600 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-secondary --no-pipeline-leg --latency-iters 0 2> $OUT/r04_c14_err.txt | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['stage_ms_per_step'], d['host_call_ms_per_step'])"
  grep TW_HOST_TIMING $OUT/r04_c14_err.txt | tail -2
done
