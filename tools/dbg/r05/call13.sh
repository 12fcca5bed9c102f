#!/bin/bash
# round 5, call 13: the driver's bench line on the final tree (wall time of the whole command), and the drop-in API at 64 sessions per GPU
O=gpurun_out/r05_call13; mkdir -p $O
S=$(date +%s); timeout 900 python bench.py > $O/bench_driver_defaults.json 2> $O/bench_driver_defaults.err; echo "bench.py (no flags): $(( $(date +%s) - S )) s wall" | tee $O/bench_wall.txt
python -c "
import json; d=json.loads(open('$O/bench_driver_defaults.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['traffic'], d['value_contract'], d['value_f16'], d['cpu_baseline']['kind'], d['config3']['p50_ms'])"
timeout 900 python bench.py --streams 64 --steps 5 --no-cpu-baseline --no-secondary --latency-iters 10 --hub-rounds 6 --no-hub-two-cohorts --hub-short-tokens 0 > $O/bench_api_b64.json 2> $O/bench_api_b64.err
python -c "
import json; d=json.loads(open('$O/bench_api_b64.json').read().strip().splitlines()[-1]); p=d['pipeline']; print('b64', d['value'], d['value_contract'], p.get('hub_request_p50_ms'), p.get('hub_mean_rows_per_pass'), p.get('hub_phase_ms_per_pass'), p.get('error'))"
