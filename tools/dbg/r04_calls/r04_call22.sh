#!/bin/bash
# round 4, call 22: arriving requests opened on a stream of their own (THEWHISPER_HUB_INTAKE_STREAM) - GPU serving tests, hub legs A/B
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
timeout 600 python -m pytest tests/test_gpu_serving.py -m gpu -x -q 2>&1 | tail -2
for V in 1 0 1 0; do
THEWHISPER_HUB_INTAKE_STREAM=$V timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --latency-iters 5 --hub-rounds 10 --hub-short-tokens 24 --no-hub-two-cohorts --hub-prefetch-cus 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); p=d['pipeline']
print('INTAKE_STREAM=$V', 'value',d['value'],'value_api',d['value_api'],'steady',d.get('value_api_steady_state'),'hub_p50',p.get('hub_request_p50_ms'),'short_p50',p.get('hub_short_request_p50_ms'), p.get('hub_phase_ms_per_pass'))"
done
