#!/bin/bash
# round 4, call 8: encoder pre-LayerNorm fold - parity, then A/B against TW_ENC_FOLD_LN=0
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
cd $ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -5
TW_DUMP_DIR=$OUT/fd timeout 1200 python -m pytest tests/test_gpu_full_depth.py -m gpu -q -s 2>&1 | grep "FULLDEPTH\|passed\|failed\|Error\|assert" > $OUT/r04_c8_full_depth.log
tail -4 $OUT/r04_c8_full_depth.log
for V in "TW_ENC_FOLD_LN=1" "TW_ENC_FOLD_LN=0"; do echo $V; env $V timeout 600 python tools/bench_encoder.py --cases 500x16,1500x16,750x64,500x1,1500x1 2>&1 | grep encode_ms; done | tee $OUT/r04_c8_enc_fold_ab.txt
cd /tmp
d=/tmp/kt; rm -rf $d
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $d -o p -- python $ROOT/tools/bench_encoder.py --cases 500x16 > /dev/null 2>&1
t=$(find $d -name "*kernel_trace.csv" | head -1)
(cd $ROOT && python tools/trace_by_shape.py $t 14 > $OUT/r04_c8_encoder_by_shape.txt)
head -12 $OUT/r04_c8_encoder_by_shape.txt
