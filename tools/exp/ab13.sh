#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
run() { env $1 python bench.py --no-cpu-baseline --no-pipeline-leg --latency-iters 0 --steps 4 --encoder-cus 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$2 no-overlap', d['value'], d['roofline']['avg_step_ms'])"; }
P=$PWD/thewhisper_amd/lib/variants/libPREV.so
run "TW_X=1" new
run "THEWHISPER_LIB=$P" prev
