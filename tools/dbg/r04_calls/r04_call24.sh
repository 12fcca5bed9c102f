#!/bin/bash
# round 4, call 24: s_setprio around the MFMA block of the large-M GEMM (TW_GEMM_PRIO = 0 / 1 / 2 / 3)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
for V in 0 1 2 3 0 1; do echo "TW_GEMM_PRIO=$V"; TW_GEMM_PRIO=$V timeout 600 python tools/bench_encoder.py --cases 500x16,1500x16,750x64 2>&1 | grep encode_ms; done
