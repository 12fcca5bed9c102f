#!/bin/bash
# round 5, call 15: where the decode loop runs, re-swept with round 5's kernels: 16 / 32 / 64 / 1 streams on 256 / 224 / 192 / 160 / 128 compute units
O=gpurun_out/r05_call15; mkdir -p $O
timeout 900 python tools/dbg/decode_cu_mask.py --cus 256,224,192,160,128 --streams 16,32,64,1 > $O/decode_cu_mask.txt 2>&1
grep "streams=" $O/decode_cu_mask.txt
