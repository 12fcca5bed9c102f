#!/bin/bash
O=gpurun_out/r05_call8; mkdir -p $O
run() { env "$@" 2>&1 | grep "mode=" >> $O/sym3.txt; }
P="timeout 300 python tools/dbg/r05/group_symmetry3.py"
run ENC_LAYERS=32 TW_SK_CG_MODE=3 $P 32 1500 1 24
run ENC_LAYERS=32 TW_SK_CG_MODE=2 $P 32 1500 1 24
run ENC_LAYERS=32 TW_SK_CG_MODE=3 $P 32 1500 0 24
run ENC_LAYERS=32 TW_SK_CG_MODE=3 THEWHISPER_DECODE_CUS=0 $P 32 1500 1 24
run ENC_LAYERS=32 TW_SK_CG_MODE=3 TW_FUSE_EMBED=0 $P 32 1500 1 24
run ENC_LAYERS=32 TW_SK_CG_MODE=3 $P 32 500 1 60
cat $O/sym3.txt
