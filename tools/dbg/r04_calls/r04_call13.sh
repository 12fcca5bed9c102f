#!/bin/bash
# round 4, call 13: deep DMA rings for the single-stream GEMM shapes - parity, A/B against the 2- / 3-stage rings, per-shape trace
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
cd $ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "encoder or enc or gemm or full or conv" 2>&1 | tail -3
for V in "TW_GEMM_DEEP=1" "TW_GEMM_DEEP=0" "TW_GEMM_CFG=10" "TW_GEMM_CFG=11" "TW_GEMM_CFG=12"; do echo $V; env $V timeout 600 python tools/bench_encoder.py --cases 500x1,1500x1,500x2,500x3,750x1 2>&1 | grep encode_ms; done | tee $OUT/r04_c13_deep_ring_ab.txt
cd /tmp
for CASE in 500x1 1500x1; do
  d=/tmp/kt_$CASE; rm -rf $d
  timeout 600 rocprofv3 --kernel-trace --output-format csv -d $d -o p -- python $ROOT/tools/bench_encoder.py --cases $CASE > /dev/null 2>&1
  t=$(find $d -name "*kernel_trace.csv" | head -1)
  (cd $ROOT && python tools/trace_by_shape.py $t 14 > $OUT/r04_c13_encoder_by_shape_$CASE.txt)
  head -9 $OUT/r04_c13_encoder_by_shape_$CASE.txt
done
