#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp; export TMPDIR=/tmp
for dt in bf16 fp8; do
  rm -rf /tmp/prof_$dt
  timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_$dt -o p -- python $ROOT/tools/bench_decode.py --layers 8 --batches 16 --T 750 --dtype $dt > /tmp/prof_$dt.log 2>&1
  t=$(find /tmp/prof_$dt -name "*kernel_trace.csv" | head -1)
  echo "=== $dt"; grep "^B=" /tmp/prof_$dt.log; (cd $ROOT && python tools/trace_by_shape.py $t 12)
done
