#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_one
rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_one -o p -- python $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-pipeline-leg --latency-iters 0 --encoder-cus 0 > /dev/null 2>&1
t=$(find /tmp/prof_one -name "*kernel_trace.csv" | head -1)
(cd $ROOT && python tools/trace_by_shape.py $t 10 > $OUT/prof_one_by_shape.txt)
