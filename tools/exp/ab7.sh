#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
run() { env $1 python bench.py --no-cpu-baseline --no-pipeline-leg --latency-iters 0 --steps 6 $2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1 $2', d['value'], d['roofline']['avg_step_ms'])"; }
for i in 1 2; do
  run "TW_X=0" ""
  run "TW_SK_MAX_BLOCKS=256" ""
  run "TW_SK_MAX_BLOCKS=300" ""
done
