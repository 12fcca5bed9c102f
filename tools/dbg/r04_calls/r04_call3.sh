#!/bin/bash
# Round 4, GPU call 3: tile-height picker + attention VALU diet (encoder A/B), decode at 64 streams with two tiles per workgroup
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
(time timeout 1500 python -m pytest tests -m gpu -q -s 2>&1) > $OUT/r04_c3_gpu_tests.log 2>&1
grep -E "passed|failed|FAILED" $OUT/r04_c3_gpu_tests.log | tail -8
for V in "TW_GEMM_BM=128" "X=1" "TW_GEMM_BM=96" "TW_GEMM_BM=80" "TW_GEMM_BM_OVH=8" "TW_GEMM_BM_OVH=48" "TW_GEMM_BM=128" "X=1"; do env $V timeout 600 python tools/bench_encoder.py --cases 500x16,750x16,1500x16,500x4,500x8,500x32; done > $OUT/r04_c3_encoder.txt 2>&1
grep encode_ms $OUT/r04_c3_encoder.txt
cd /tmp
d=/tmp/kt_enc; rm -rf $d
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $d -o p -- python $ROOT/tools/bench_encoder.py --cases 500x16 > /dev/null 2>&1
t=$(find $d -name "*kernel_trace.csv" | head -1)
(cd $ROOT && python tools/trace_by_shape.py $t 12 > $OUT/r04_c3_encoder_by_shape.txt)
cat $OUT/r04_c3_encoder_by_shape.txt | head -8
cd $ROOT
A="--steps 6 --warmup 2 --no-cpu-baseline --no-secondary --no-pipeline-leg --latency-iters 0 --chunk-s 15 --streams 64"
for V in "X=1" "TW_SK_MAX_BLOCKS=160" "X=1" "TW_SK_MAX_BLOCKS=160"; do
  for DT in bf16 fp8a16; do
    env $V timeout 900 python bench.py $A --dtype $DT > $OUT/r04_c3_tmp.json 2>/dev/null
    python -c "
import json; d=json.load(open('$OUT/r04_c3_tmp.json')); print('$V', '$DT', 64, 'tok/s', d['value'], 'step_ms', d['roofline']['avg_step_ms'])"
  done
done | tee $OUT/r04_c3_b64_two_tiles.txt
timeout 900 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --no-pipeline-leg --latency-iters 20 > $OUT/r04_c3_bench_quick.json 2> $OUT/r04_c3_bench_quick.err
python -c "
import json; d=json.load(open('$OUT/r04_c3_bench_quick.json')); print('headline', d['value'], d['roofline']['avg_step_ms'], d['ms_per_step'], d.get('stage_ms_per_step'), d.get('p50_chunk_latency_ms'), d.get('streaming_tick_breakdown'))"
