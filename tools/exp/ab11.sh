#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
run() { env $1 python bench.py --no-cpu-baseline --no-pipeline-leg --latency-iters 0 --steps 6 $2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$3 $2', d['value'], d['roofline']['avg_step_ms'])"; }
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "teacher_forced or greedy or cross_query" 2>&1 | tail -3
P=$PWD/thewhisper_amd/lib/variants/libPREV.so
for i in 1 2; do
  run "TW_X=1" "" new
  run "THEWHISPER_LIB=$P" "" prev
done
run "TW_X=1" "--model large-v3-turbo --chunk-s 30 --streams 1 --encoder-cus 0" new
run "THEWHISPER_LIB=$P" "--model large-v3-turbo --chunk-s 30 --streams 1 --encoder-cus 0" prev
run "TW_X=1" "--dtype fp8 --chunk-s 15" new
run "THEWHISPER_LIB=$P" "--dtype fp8 --chunk-s 15" prev
