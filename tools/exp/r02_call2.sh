#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
( timeout 900 python bench.py > $OUT/c2_bench_default.json 2> $OUT/c2_bench_default.err ); echo "bench rc=$?" > $OUT/c2_status.log
( timeout 600 python -m pytest tests/test_gpu_full_depth.py -q -m gpu -s -k "large-v3_c10-bf16" 2>&1 | grep -E "FULLDEPTH|passed|failed|Error" | head -20 ) > $OUT/c2_fulldepth.log
tail -c 1500 $OUT/c2_bench_default.err
