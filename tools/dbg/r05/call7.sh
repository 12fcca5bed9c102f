#!/bin/bash
O=gpurun_out/r05_call7; mkdir -p $O
run() { env "$@" 2>&1 | grep "mode=" >> $O/sym3.txt; }
P="timeout 300 python tools/dbg/r05/group_symmetry3.py"
run TW_SK_CG_MODE=3 $P 32 1500 1 24
run TW_SK_CG_MODE=2 $P 32 1500 1 24
run TW_SK_CG_MODE=3 $P 32 100 1 60
run TW_SK_CG_MODE=3 $P 32 100 0 60
run TW_SK_CG_MODE=3 THEWHISPER_DECODE_CUS=0 $P 32 100 1 60
run TW_SK_CG_MODE=3 $P 8 100 1 100
run TW_SK_CG_MODE=3 $P 2 100 1 100
run TW_SK_CG_MODE=2 $P 32 100 1 60
cat $O/sym3.txt
