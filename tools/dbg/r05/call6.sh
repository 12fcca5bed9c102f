#!/bin/bash
O=gpurun_out/r05_call6; mkdir -p $O
for L in 1 2; do for m in 2 3; do TW_SK_CG_MODE=$m timeout 300 python tools/dbg/r05/group_symmetry2.py $L 2>&1 | grep "mode=" >> $O/sym2.txt; done; done
cat $O/sym2.txt
