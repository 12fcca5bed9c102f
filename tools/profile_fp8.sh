#!/bin/bash
# Per-kernel comparison of the fp8 context with bf16 at BASELINE config 5's shape (15 s chunks, 16 streams): rocprofv3 kernel trace
# of one timed step each, no encoder overlap, grouped by (kernel, grid).   bash tools/profile_fp8.sh r03
set -u
R=${1:-r03}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--chunk-s 15 --steps 1 --warmup 1 --no-cpu-baseline --no-pipeline-leg --no-secondary --latency-iters 0 --encoder-cus 0"
for DT in fp8 bf16; do
  d=/tmp/prof_$DT; rm -rf $d
  rocprofv3 --kernel-trace --output-format csv -d $d -o p -- python $ROOT/bench.py --dtype $DT $ARGS > $OUT/${R}_bench_${DT}_15s_profiled.json 2>/dev/null
  t=$(find $d -name "*kernel_trace.csv" | head -1)
  (cd $ROOT && python tools/trace_by_shape.py $t 16 > $OUT/${R}_${DT}_15s_by_shape.txt)
done
