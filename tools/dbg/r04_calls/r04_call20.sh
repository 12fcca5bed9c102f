#!/bin/bash
# round 4, call 20: projection grids capped at 160 / 320 workgroups (several tiles per workgroup) against the default 512
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
for V in 512 160 320 512 160; do
TW_SK_MAX_BLOCKS=$V timeout 600 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-secondary --no-pipeline-leg --latency-iters 20 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('TW_SK_MAX_BLOCKS=$V', 'value',d['value'],'ms/step',d['ms_per_step'],'avg_step_ms',d['roofline']['avg_step_ms'],'p50_chunk',d['p50_chunk_latency_ms'])"
done
