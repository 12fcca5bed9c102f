#!/bin/bash
# round 5, call 16: the float16 id-identity assertion of the full-depth suite
O=gpurun_out/r05_call16; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_full_depth.py -m gpu -x -q -k "f16" 2>&1 | tail -3 | tee $O/f16_ids.txt
