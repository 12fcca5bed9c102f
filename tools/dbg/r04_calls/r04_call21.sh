#!/bin/bash
# round 4, call 21: workgroups of the logits launch (TW_SK_MAX_BLOCKS_TALL)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
for V in 320 512 160 240 480 320 512; do
TW_SK_MAX_BLOCKS_TALL=$V timeout 600 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-secondary --no-pipeline-leg --latency-iters 20 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('TW_SK_MAX_BLOCKS_TALL=$V', 'value',d['value'],'ms/step',d['ms_per_step'],'avg_step_ms',d['roofline']['avg_step_ms'],'p50_chunk',d['p50_chunk_latency_ms'])"
done
