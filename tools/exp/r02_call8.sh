#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
rm -f $OUT/c8_*
for c in -1 5 6 4 1 0; do ( TW_GEMM_CFG=$c timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "test_encoder or decoder_teacher_forced and large" 2>&1 | tail -3 ) > $OUT/c8_parity_cfg$c.log; done
( TW_TORCH_REF=1 timeout 600 python tools/bench_encoder.py 2>&1 | grep "^T=\|^torch" ) >> $OUT/c8_encoder_variants.log
for cfg in "TW_GEMM_CFG=6" "TW_GEMM_CFG=4" "TW_GEMM_CFG=1" "TW_GEMM_CFG=0"; do
  ( env $cfg timeout 600 python tools/bench_encoder.py 2>&1 | grep "^T=" ) >> $OUT/c8_encoder_variants.log
done
cd /tmp; rm -rf /tmp/prof_enc
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_enc -o enc -- python $ROOT/tools/bench_encoder.py --cases 500x16 > /tmp/prof_enc.log 2>&1
t=$(find /tmp/prof_enc -name "*kernel_trace.csv" | head -1)
(cd $ROOT && python tools/trace_by_shape.py $t 14 > $OUT/c8_enc_byshape.txt)
tail -n 2 $OUT/c8_parity_cfg*.log | cut -c1-200
cat $OUT/c8_encoder_variants.log $OUT/c8_enc_byshape.txt
