#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_e
rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_e -o p -- python $ROOT/tools/bench_encoder.py --cases 1500x1,500x1 > $OUT/enc1500.log 2>&1
t=$(find /tmp/prof_e -name "*kernel_trace.csv" | head -1)
(cd $ROOT && python tools/trace_by_shape.py $t 24 > $OUT/enc1500_by_shape.txt)
tail -3 $OUT/enc1500.log
