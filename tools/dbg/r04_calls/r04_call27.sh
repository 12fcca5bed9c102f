#!/bin/bash
# round 4, call 27: the V^T segment of the QKV projection staged too (transposed in LDS, 8-byte stores) - parity, encoder timing
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "encoder or enc or gemm or full or conv" 2>&1 | tail -2
timeout 600 python -m pytest tests/test_gpu_full_depth.py -m gpu -x -q -k "c10-bf16 or c15-f16 or turbo_c30-bf16 or c10_b16-f16" 2>&1 | tail -2
for V in 1 1; do echo "TW_GEMM_STAGED=$V"; TW_GEMM_STAGED=$V timeout 300 python tools/bench_encoder.py --cases 500x16,1500x16,750x64,500x4 2>&1 | grep encode_ms | cut -c1-120; done | tee $OUT/r04_c27_staged_vt.txt
