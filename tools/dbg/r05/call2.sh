#!/bin/bash
# round 5, call 2: the projection kernel for more than 16 streams (operand rings, padded LDS tiles, weights-first order) - standalone probe
# of the five projection launches of a layer on 160 CUs (what the decode loop gets), then the GPU suite and the decode step itself
set -x
O=gpurun_out/r05_call2; mkdir -p $O
cd tools/dbg
for m in 0; do PROBE_CUS=160 TW_SK_CG_MODE=$m ./probe_gemv_r04 > ../../$O/probe_r04_mode$m.txt 2>&1; done
for m in 0 1 2; do PROBE_CUS=160 TW_SK_CG_MODE=$m ./probe_gemv_new > ../../$O/probe_new_mode$m.txt 2>&1; done
PROBE_CUS=160 TW_SK_CG_MODE=2 ./probe_gemv_new_noeall > ../../$O/probe_new_noeall_mode2.txt 2>&1
TW_SK_CG_MODE=2 ./probe_gemv_new > ../../$O/probe_new_mode2_allcus.txt 2>&1
TW_SK_CG_MODE=0 ./probe_gemv_r04 > ../../$O/probe_r04_mode0_allcus.txt 2>&1
PROBE_CUS=160 PROBE_B=64 TW_SK_CG_MODE=0 ./probe_gemv_r04_ts > ../../$O/probe_r04_ts_b64.txt 2>&1
PROBE_CUS=160 PROBE_B=64 TW_SK_CG_MODE=2 ./probe_gemv_new_ts > ../../$O/probe_new_ts_b64.txt 2>&1
PROBE_CUS=160 PROBE_B=16 TW_SK_CG_MODE=0 ./probe_gemv_r04_ts > ../../$O/probe_r04_ts_b16.txt 2>&1
PROBE_CUS=160 PROBE_B=16 TW_SK_CG_MODE=2 ./probe_gemv_new_ts > ../../$O/probe_new_ts_b16.txt 2>&1
cd ../..
tail -20 $O/probe_r04_mode0.txt; tail -20 $O/probe_new_mode2.txt
timeout 900 python -m pytest tests -m gpu -x -q -s > $O/gpu_tests.log 2>&1; echo "pytest rc=$?" >> $O/gpu_tests.log
tail -5 $O/gpu_tests.log
for m in 2 1 0; do TW_SK_CG_MODE=$m timeout 300 python tools/bench_decode.py --layers 32 --batches 16,32,64 --tokens 128 > $O/decode_mode$m.txt 2>&1; tail -3 $O/decode_mode$m.txt; done
