#!/bin/bash
O=gpurun_out/r05_call9; mkdir -p $O
run() { env "$@" 2>&1 | grep "mode=" >> $O/sym3.txt; }
P="timeout 300 python tools/dbg/r05/group_symmetry3.py"
run ENC_LAYERS=32 TW_SK_CG_MODE=3 $P 32 500 1 60
run ENC_LAYERS=32 TW_SK_CG_MODE=3 TW_SK_DUAL_WHICH=1 $P 32 500 1 60
run ENC_LAYERS=32 TW_SK_CG_MODE=3 TW_SK_DUAL_WHICH=2 $P 32 500 1 60
run ENC_LAYERS=32 TW_SK_CG_MODE=3 THEWHISPER_LIB=$PWD/tools/dbg/libtw_noeall.so $P 32 500 1 60
cat $O/sym3.txt
cd tools/dbg
for v in 0 8 1; do PROBE_CUS=160 PROBE_B=16 TW_SK_DUAL16_MIN_B=$v ./probe_gemv_new > ../../$O/probe_b16_dual16_$v.txt 2>&1; done
for B in 1 4 8; do for v in 0 1; do PROBE_CUS=160 PROBE_B=$B TW_SK_DUAL16_MIN_B=$v ./probe_gemv_new > ../../$O/probe_b${B}_dual16_$v.txt 2>&1; done; done
PROBE_CUS=160 PROBE_B=16 TW_SK_DUAL16_MIN_B=8 ./probe_gemv_new_ts > ../../$O/probe_ts_b16_dual16.txt 2>&1
cd ../..
grep -H "sum of" $O/probe_b*.txt
for v in 0 1; do TW_SK_DUAL16_MIN_B=$v timeout 300 python tools/bench_decode.py --layers 32 --batches 1,4,8,16 --tokens 128 > $O/decode_dual16_$v.txt 2>&1; grep "B=" $O/decode_dual16_$v.txt; done
