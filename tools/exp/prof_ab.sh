#!/bin/bash
# per-launch durations of the decode step, fused vs unfused cross query (same box)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for g in 1 0; do
  rm -rf /tmp/prof_ab$g
  TW_FUSE_CQ=$g rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_ab$g -o p -- python $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-pipeline-leg --latency-iters 0 --encoder-cus 0 $* > /dev/null 2>&1
  t=$(find /tmp/prof_ab$g -name "*kernel_trace.csv" | head -1)
  (cd $ROOT && python tools/trace_by_shape.py $t 16 > $OUT/ab_fuse${g}_by_shape.txt)
done
