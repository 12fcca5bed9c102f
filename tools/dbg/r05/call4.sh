#!/bin/bash
# round 5, call 4: which of the two changes of call 3 (two tiles per workgroup / embedding written by the sampler) breaks
# "the same audio in another group of 16 streams gives the same ids" at 64 streams
O=gpurun_out/r05_call4; mkdir -p $O
for S in "TW_SK_CG_MODE=2" "TW_FUSE_EMBED=0" "TW_SK_CG_MODE=3"; do
  echo "== $S" >> $O/which.txt
  env $S timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "maximum_context" 2>&1 | tail -3 >> $O/which.txt
done
cat $O/which.txt
