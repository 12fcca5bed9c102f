#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
run() { env $1 python bench.py --no-cpu-baseline --no-pipeline-leg --latency-iters 0 --steps 6 $2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1 $2', d['value'], d['roofline']['avg_step_ms'])"; }
for i in 1 2; do
  run "TW_X=base" ""
  run "TW_SK_NW_BIGK=8" ""
  run "TW_SK_MAX_BLOCKS=1024" ""
  run "TW_SK_MAX_BLOCKS=256" ""
  run "TW_X=cus48" "--encoder-cus 48"
  run "TW_X=cus80" "--encoder-cus 80"
done
