#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
run() { env $1 python bench.py --no-cpu-baseline --no-pipeline-leg --latency-iters 0 --steps 6 $2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1 $2', d['value'], d['roofline']['avg_step_ms'])"; }
TW_GRAPH_STEPS=4 timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "greedy_ids or eos or suppress" 2>&1 | tail -2
for i in 1 2; do
  for g in 1 2 4; do
    run "TW_GRAPH_STEPS=$g" "--model large-v3-turbo --chunk-s 30 --streams 1 --encoder-cus 0"
    run "TW_GRAPH_STEPS=$g" ""
  done
done
