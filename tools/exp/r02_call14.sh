#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out; mkdir -p $OUT
cd $ROOT
timeout 600 python -m pytest tests/test_gpu_serving.py -q -m gpu -x 2>&1 | tail -2
python bench.py > $OUT/r02_bench_default.json 2> $OUT/r02_bench_default.err
python -c "
import json
d=json.loads(open('$OUT/r02_bench_default.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['avg_step_ms'], d['roofline']['frac'], json.dumps({k:v for k,v in d.get('pipeline',{}).items() if k!='note'}))
"
tail -3 $OUT/r02_bench_default.err
