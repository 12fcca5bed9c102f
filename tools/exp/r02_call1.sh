#!/bin/bash
# round-2 GPU call 1: parity (incl. full depth) + decode-step variants (narrow tiles, cross-K/V prefetch)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -15 ) > $OUT/c1_parity.log
( TW_SK_TR=4 timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "decoder_teacher or greedy_ids" 2>&1 | tail -8 ) > $OUT/c1_parity_tr4.log
( TW_PF_BLOCKS=64 TW_PF_MIN_B=1 timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "greedy_ids or full_size or overlap" 2>&1 | tail -8 ) > $OUT/c1_parity_pf.log
( timeout 1500 python -m pytest tests/test_gpu_full_depth.py -q -m gpu -s 2>&1 | grep -E "FULLDEPTH|passed|failed|Error|assert" | head -60 ) > $OUT/c1_fulldepth.log
for cfg in "TW_SK_TR=16" "TW_SK_TR=8" "TW_SK_TR=4" "TW_SK_TR=8 TW_PF_BLOCKS=64" "TW_SK_TR=8 TW_PF_BLOCKS=128" "TW_SK_TR=8 TW_PF_BLOCKS=256" "TW_SK_TR=16 TW_PF_BLOCKS=128"; do
  ( env $cfg timeout 300 python tools/bench_decode.py --layers 8 --batches 1,16 2>&1 | grep "^B=" ) >> $OUT/c1_decode_variants.log
done
cd /tmp
for cfg in "TW_SK_TR=16" "TW_SK_TR=8 TW_PF_BLOCKS=128"; do
  tag=$(echo $cfg | tr ' =' '__')
  rm -rf /tmp/prof_dec
  env $cfg timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_dec -o dec -- python $ROOT/tools/bench_decode.py --layers 8 --batches 16 > /tmp/prof_dec.log 2>&1
  t=$(find /tmp/prof_dec -name "*kernel_trace.csv" | head -1)
  (cd $ROOT && python tools/trace_by_shape.py $t 16 > $OUT/c1_byshape_$tag.txt)
done
ls -la $OUT
