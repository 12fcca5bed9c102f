#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
rm -f $OUT/c7_*
for c in -1 5 6 4 1 0; do ( TW_GEMM_CFG=$c timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "test_encoder or decoder_teacher_forced and large" 2>&1 | tail -3 ) > $OUT/c7_parity_cfg$c.log; done
for cfg in "TW_X=auto" "TW_GEMM_CFG=5" "TW_GEMM_CFG=6" "TW_GEMM_CFG=4"; do
  ( env $cfg timeout 600 python tools/bench_encoder.py 2>&1 | grep "^T=" ) >> $OUT/c7_encoder_variants.log
done
tail -n 3 $OUT/c7_parity_cfg*.log | cut -c1-200
cat $OUT/c7_encoder_variants.log
