#!/bin/bash
# round 4, call 25: what bounds the large-M GEMM - the same kernel with parts compiled out (libraries built with -DTW_GEMM_DBG=bits from a
# scratch copy of k_gemm.hip: 1 no epilogue, 2 weight fragments loaded once, 4 no activation DMA, 8 no LDS fragment reads; results are wrong
# by construction, only the time is read) - encoder of 16 x 10 s / 16 x 30 s under rocprofv3, per GEMM shape
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
for D in 0 1 2 4 12 14 15; do
  d=/tmp/kt_$D; rm -rf $d
  THEWHISPER_LIB=$ROOT/thewhisper_amd/lib/dbg/libtw_gemm_dbg$D.so timeout 300 rocprofv3 --kernel-trace --output-format csv -d $d -o p -- python $ROOT/tools/bench_encoder.py --cases 500x16 2>&1 | grep encode_ms | cut -c1-60
  t=$(find $d -name "*kernel_trace.csv" | head -1)
  echo "== TW_GEMM_DBG=$D"; (cd $ROOT && python tools/trace_by_shape.py $t 6 | grep gemm_wreg | cut -c1-150)
done 2>&1 | tee $OUT/r04_c25_gemm_parts.txt
