#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
run() { env $1 python bench.py --no-cpu-baseline --no-pipeline-leg --latency-iters 0 --steps 5 $2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$3 $2', d['value'], d['roofline']['avg_step_ms'])"; }
timeout 200 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "cross_query or mxfp8 or greedy_ids" 2>&1 | tail -2
P=$PWD/thewhisper_amd/lib/variants/libPREV.so
run "TW_X=1" "" new
run "THEWHISPER_LIB=$P" "" prev
run "TW_X=1" "" new
