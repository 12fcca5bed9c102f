#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
run() { THEWHISPER_LIB=$1 python bench.py --no-cpu-baseline --no-pipeline-leg --latency-iters 0 --steps 6 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$2', d['value'], d['roofline']['avg_step_ms'])"; }
for i in 1 2 3; do
  run $PWD/thewhisper_amd/lib/variants/libA.so A_head
  run $PWD/thewhisper_amd/lib/variants/libB.so B_selfattn_nw
  run $PWD/thewhisper_amd/lib/libthewhisper_gfx950.so C_all
done
