#!/bin/bash
# Round 4, GPU call 1 (run through gpurun from the repo root): parity suite, encoder A/B of the XCD-aware tile order, fp8 flavours.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
(time timeout 1200 python -m pytest tests -m gpu -q -x -s 2>&1) > $OUT/r04_c1_gpu_tests.log 2>&1
tail -5 $OUT/r04_c1_gpu_tests.log
# encoder A/B (GPU to itself)
: > $OUT/r04_c1_encoder_ab.txt
for V in "TW_GEMM_XCD=0 TW_ATTN_XCD=0" "TW_GEMM_XCD=1 TW_ATTN_XCD=0" "TW_GEMM_XCD=0 TW_ATTN_XCD=1" "TW_GEMM_XCD=1 TW_ATTN_XCD=1"; do
  env $V timeout 600 python tools/bench_encoder.py --cases 500x16,1500x1,500x1,750x16,1500x16 >> $OUT/r04_c1_encoder_ab.txt 2>&1
done
cat $OUT/r04_c1_encoder_ab.txt | grep encode_ms
# fp8 flavours at config 5's shape and at 32 / 64 streams
A="--steps 6 --warmup 2 --no-cpu-baseline --no-secondary --no-pipeline-leg --latency-iters 0 --chunk-s 15"
: > $OUT/r04_c1_fp8_ab.txt
for S in 16 32 64; do
  for DT in bf16 fp8a8 fp8a16; do
    if [ "$DT" = "fp8a8" ] && [ $S -gt 16 ]; then continue; fi
    timeout 900 python bench.py $A --streams $S --dtype $DT > $OUT/r04_c1_fp8_${DT}_${S}.json 2> $OUT/r04_c1_fp8_${DT}_${S}.err
    python - <<PY >> $OUT/r04_c1_fp8_ab.txt
import json
try:
    d = json.load(open("$OUT/r04_c1_fp8_${DT}_${S}.json"))
    print("$DT", $S, "tok/s", d["value"], "step_ms", d["roofline"]["avg_step_ms"], "ms_per_step", d["ms_per_step"], "enc", d.get("stage_ms_per_step"))
except Exception as e:
    print("$DT", $S, "FAILED", e)
PY
  done
done
cat $OUT/r04_c1_fp8_ab.txt
# headline regression check
timeout 900 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --no-pipeline-leg --latency-iters 0 > $OUT/r04_c1_bench_quick.json 2> $OUT/r04_c1_bench_quick.err
python -c "
import json; d=json.load(open('$OUT/r04_c1_bench_quick.json')); print('headline', d['value'], d['roofline']['avg_step_ms'], d['ms_per_step'], d.get("stage_ms_per_step"))"
