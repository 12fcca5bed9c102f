#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_parity.py -x -v -m gpu -k "encoder or decoder_teacher or pipeline_on_gpu" 2>&1 | grep -v "^  File\|Extension modules" | head -80 ) > $OUT/c4_parity.log
cat $OUT/c4_parity.log | cut -c1-300
