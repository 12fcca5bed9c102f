#!/bin/bash
O=gpurun_out/r05_call5; mkdir -p $O
for m in 2 3; do for B in 64 32; do TW_SK_CG_MODE=$m timeout 300 python tools/dbg/r05/group_symmetry.py $B 1 2>&1 | grep "mode=" >> $O/sym.txt; done; done
cat $O/sym.txt
