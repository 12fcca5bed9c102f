#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
run() { python bench.py --no-cpu-baseline --no-pipeline-leg --latency-iters 0 --steps 5 --encoder-cus $1 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('encoder-cus $1', d['value'], d['ms_per_step'], d['roofline']['avg_step_ms'])"; }
run 64
run 96
run 64
run 96
