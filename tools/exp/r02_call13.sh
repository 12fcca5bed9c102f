#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out; mkdir -p $OUT
cd $ROOT
python bench.py > $OUT/r02_bench_default.json 2>/dev/null
for s in 1 8 32 64; do
  python bench.py --streams $s --no-cpu-baseline --no-pipeline-leg --latency-iters 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('streams $s', d['value'], d['roofline']['avg_step_ms'], d['roofline']['frac'])"
done | tee $OUT/r02_streams_sweep.txt
python bench.py --model large-v3-turbo --chunk-s 30 --streams 1 --no-cpu-baseline --latency-iters 30 > $OUT/r02_bench_turbo_30s_b1.json 2>/dev/null
python -c "
import json
for f in ('r02_bench_default.json','r02_bench_turbo_30s_b1.json'):
    d=json.loads(open('$OUT/'+f).read().strip().splitlines()[-1])
    print(f, d['value'], d['ms_per_step'], d['roofline']['avg_step_ms'], d['roofline']['frac'], d['roofline'].get('traffic'), d.get('pipeline',{}).get('hub_tok_per_s'), d.get('pipeline',{}).get('backend_transcribe_p50_ms'), d.get('pipeline',{}).get('scheduler_pattern_p50_ms'), d.get('p50_chunk_latency_ms'))
"
