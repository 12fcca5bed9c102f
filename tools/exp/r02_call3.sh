#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "encoder or decoder_teacher or pipeline_on_gpu" 2>&1 | tail -6 ) > $OUT/c3_parity.log
for c in 0 1 2 3; do ( TW_GEMM_CFG=$c timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "test_encoder and bf16" 2>&1 | tail -3 ) > $OUT/c3_parity_cfg$c.log; done
for cfg in "TW_X=auto" "TW_GEMM_CFG=4" "TW_GEMM_CFG=2" "TW_GEMM_CFG=3" "TW_GEMM_CFG=1" "TW_GEMM_CFG=0"; do
  ( env $cfg timeout 600 python tools/bench_encoder.py 2>&1 | grep "^T=" ) >> $OUT/c3_encoder_variants.log
done
cat $OUT/c3_encoder_variants.log
