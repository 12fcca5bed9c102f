#!/bin/bash
# round 4, call 16: greedy loop on its own 160-CU stream by default - whole GPU suite, driver line (hub + latency legs), 32 / 64 streams
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
timeout 900 python tools/dbg_decode_cu_mask.py --cus 256,160 --streams 64,32,4 2>&1 | grep "device loop" | tee $OUT/r04_c16_decode_cu_mask_64.txt
for V in 160 0; do
THEWHISPER_DECODE_CUS=$V timeout 900 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-secondary --latency-iters 30 --hub-rounds 8 --hub-short-tokens 0 --no-hub-two-cohorts --hub-prefetch-cus 0 2> $OUT/r04_c16_err.txt | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); p=d['pipeline']
print('DECODE_CUS=$V', 'value',d['value'],'value_api',d['value_api'],'p50_chunk',d['p50_chunk_latency_ms'],'backend_p50',p.get('backend_transcribe_p50_ms'),'sched_p50',p.get('scheduler_pattern_p50_ms'),'hub_p50',p.get('hub_request_p50_ms'), p.get('hub_phase_ms_per_pass'))"
done
