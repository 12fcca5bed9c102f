#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
timeout 900 python tools/dbg_decode_cu_mask.py 2>&1 | grep "device loop" | tee $OUT/r04_c15_decode_cu_mask.txt
