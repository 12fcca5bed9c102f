#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
run() { env $1 python bench.py --no-cpu-baseline --no-pipeline-leg --latency-iters 0 --steps 6 $2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1 $2', d['value'], d['roofline']['avg_step_ms'])"; }
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -15
for i in 1 2; do
  for g in 1 0; do
    run "TW_FUSE_CQ=$g" ""
    run "TW_FUSE_CQ=$g" "--model large-v3-turbo --chunk-s 30 --streams 1 --encoder-cus 0"
  done
done
run "TW_FUSE_CQ=1" "--dtype fp8 --chunk-s 15"
run "TW_FUSE_CQ=0" "--dtype fp8 --chunk-s 15"
