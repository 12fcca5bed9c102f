#!/bin/bash
# round 4, call 28: same-box A/B of the staged epilogue: 1 = rows + V^T, 2 = rows only, 0 = off
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
for V in 1 2 0 1 2; do echo "TW_GEMM_STAGED=$V"; TW_GEMM_STAGED=$V timeout 300 python tools/bench_encoder.py --cases 500x16,1500x16,750x64 2>&1 | grep encode_ms | cut -c1-120; done | tee $OUT/r04_c28_staged_vt_ab.txt
