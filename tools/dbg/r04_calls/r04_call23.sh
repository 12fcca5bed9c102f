#!/bin/bash
# round 4, call 23: CU partition of the overlap pipeline with the decode loop fixed at 160 CUs (encoder 96 / 80 / 64 / 48), and 144 / 176
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
for V in "96 0" "64 160" "80 160" "48 160" "96 0" "64 160" "112 144" "80 176"; do
set -- $V
timeout 600 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-secondary --no-pipeline-leg --latency-iters 0 --encoder-cus $1 --decoder-cus $2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('enc/dec CUs $1 / $2', 'value',d['value'],'ms/step',d['ms_per_step'],'avg_step_ms',d['roofline']['avg_step_ms'], d['stage_ms_per_step'])"
done
