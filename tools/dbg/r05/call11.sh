#!/bin/bash
# round 5, call 11: deep LDS rings for the NARROW projections of a single stream only (TW_GEMM_DEEP_NARROW bit 0: 64 x 64 x 8 stages at M <= 1000,
# bit 1: 128 x 64 x 6 stages above) - encoder parity, then the encoder stage per setting
O=gpurun_out/r05_call11; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "encoder" 2>&1 | tail -2 > $O/enc_parity.txt; cat $O/enc_parity.txt
for v in 0 1 2 3 0 3; do TW_GEMM_DEEP_NARROW=$v timeout 300 python tools/bench_encoder.py --cases 500x1,750x1,500x2,1500x1,500x3 2>&1 | grep "T=" >> $O/encoder_deep_narrow.txt; done
cat $O/encoder_deep_narrow.txt
