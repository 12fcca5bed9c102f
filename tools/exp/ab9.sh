#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
run() { env $1 python bench.py --no-cpu-baseline --no-pipeline-leg --latency-iters 0 --steps 6 $2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1 $2', d['value'], d['roofline']['avg_step_ms'])"; }
TW_SK_TR=5 timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "teacher_forced and not mxfp8" 2>&1 | tail -2
for i in 1 2; do
  run "TW_SK_TR=8" ""
  run "TW_SK_TR=5" ""
done
