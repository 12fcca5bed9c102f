#!/bin/bash
# round 4, call 19: does the 160-CU decode stream cost the SHORT generate calls of the reuse path anything?
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
for V in 160 0 160 0; do
THEWHISPER_DECODE_CUS=$V timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --latency-iters 20 --hub-rounds 1 --hub-short-tokens 0 --no-hub-two-cohorts --hub-prefetch-cus 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); p=d['pipeline']
print('DECODE_CUS=$V', 'p50_chunk',d['p50_chunk_latency_ms'],'sched_p50',p.get('scheduler_pattern_p50_ms'),'reuse_p50',p.get('reuse_scheduler_pattern_p50_ms'),'reuse_p90',p.get('reuse_scheduler_pattern_p90_ms'),'backend_p50',p.get('backend_transcribe_p50_ms'))"
done
