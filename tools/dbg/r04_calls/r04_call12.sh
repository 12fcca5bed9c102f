#!/bin/bash
# round 4, call 12: serving with two cohorts (rows that sit a pass out are encoded on the side stream) + GPU serving tests
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
timeout 600 python -m pytest tests/test_gpu_serving.py -m gpu -x -q 2>&1 | tail -2

timeout 900 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-secondary --latency-iters 10 --hub-rounds 8 --hub-short-tokens 0 > $OUT/r04_c12_bench.json 2> $OUT/r04_c12_bench.err
tail -3 $OUT/r04_c12_bench.err
python - <<'PY'
import json,os
d=json.loads(open(os.path.join(os.environ.get("GRAFT_REPO_ROOT","."),"gpurun_out/r04_c12_bench.json")).read().strip().splitlines()[-1])
p=d.get("pipeline",{})
print("value",d["value"],"ms/step",d["ms_per_step"],"value_api",d.get("value_api"),"two_cohorts",d.get("value_api_two_cohorts"))
for k,v in p.items():
    if k.startswith("hub_") and ("tok_per_s" in k or "phase" in k or "p50" in k): print(k,v)
print(d.get("stage_ms_per_step"))
PY
