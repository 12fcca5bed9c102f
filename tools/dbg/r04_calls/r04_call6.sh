#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
for CASE in 500x1 1500x1; do
  d=/tmp/kt_$CASE; rm -rf $d
  timeout 600 rocprofv3 --kernel-trace --output-format csv -d $d -o p -- python $ROOT/tools/bench_encoder.py --cases $CASE > $OUT/r04_c6_enc_$CASE.txt 2>&1
  t=$(find $d -name "*kernel_trace.csv" | head -1)
  (cd $ROOT && python tools/trace_by_shape.py $t 14 > $OUT/r04_c6_encoder_by_shape_$CASE.txt)
  grep encode_ms $OUT/r04_c6_enc_$CASE.txt
  head -9 $OUT/r04_c6_encoder_by_shape_$CASE.txt
done
cd $ROOT
for V in "X=1" "TW_GEMM_CFG=0" "TW_GEMM_WREG_MIN=200"; do env $V timeout 600 python tools/bench_encoder.py --cases 500x1,1500x1,500x2,500x3; done 2>&1 | grep encode_ms
