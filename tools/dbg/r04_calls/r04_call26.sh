#!/bin/bash
# round 4, call 26: the large-M GEMM's epilogue staged through LDS (full-line stores) - parity, A/B against TW_GEMM_STAGED=0
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "encoder or enc or gemm or full or conv" 2>&1 | tail -3
timeout 600 python -m pytest tests/test_gpu_full_depth.py -m gpu -x -q 2>&1 | tail -3
for V in 1 0 1 0; do echo "TW_GEMM_STAGED=$V"; TW_GEMM_STAGED=$V timeout 300 python tools/bench_encoder.py --cases 500x16,1500x16,750x64,500x4 2>&1 | grep encode_ms | cut -c1-120; done | tee $OUT/r04_c26_staged_epilogue_ab.txt
