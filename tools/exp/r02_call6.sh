#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
( timeout 600 python -m pytest tests/test_vad.py tests/test_gpu_serving.py -x -q -m gpu 2>&1 | tail -25 ) > $OUT/c6_newtests.log
( timeout 900 python benchmark/run_rtfx.py 2>&1 | tail -3 ) > $OUT/c6_rtfx.log
( timeout 1200 python -m pytest tests -q -m gpu 2>&1 | tail -8 ) > $OUT/c6_allgpu.log
cat $OUT/c6_newtests.log $OUT/c6_rtfx.log $OUT/c6_allgpu.log | cut -c1-400
