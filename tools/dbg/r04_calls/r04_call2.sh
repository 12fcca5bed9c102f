#!/bin/bash
# Round 4, GPU call 2: full parity suite with dumps, TW_FUSE_CQ=0 surfaces, PMC counters of the encoder GEMMs, decode by shape at 64 streams,
# the hub leg with / without prefetch.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
(time TW_DUMP_DIR=$OUT/dump_r04 timeout 1500 python -m pytest tests -m gpu -q -s 2>&1) > $OUT/r04_c2_gpu_tests.log 2>&1
tail -4 $OUT/r04_c2_gpu_tests.log
(TW_FUSE_CQ=0 TW_DUMP_DIR=$OUT/dump_r04_nofuse timeout 600 python -m pytest tests/test_gpu_full_depth.py -m gpu -q -s -k "c10 and bf16" 2>&1) > $OUT/r04_c2_nofuse.log 2>&1
tail -2 $OUT/r04_c2_nofuse.log
# encoder: final dispatcher vs plain
for V in "TW_GEMM_XCD=0 TW_ATTN_XCD=0" "X=1"; do env $V timeout 600 python tools/bench_encoder.py --cases 500x16,1500x1,500x1,750x16; done > $OUT/r04_c2_encoder.txt 2>&1
grep encode_ms $OUT/r04_c2_encoder.txt
cd /tmp
rocprofv3 -L > $OUT/r04_c2_counters_list.txt 2>&1
P="python $ROOT/tools/bench_encoder.py --cases 500x16"
for SET in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE" "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES"; do
  tag=$(echo $SET | cut -d' ' -f1)
  d=/tmp/pmc_$tag; rm -rf $d
  timeout 600 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d $d -o p -- $P > /dev/null 2> $OUT/r04_c2_pmc_$tag.err
  c=$(find $d -name "*counter_collection.csv" | head -1)
  if [ -n "$c" ]; then (cd $ROOT && python tools/pmc_table.py $c gemm > $OUT/r04_c2_pmc_$tag.txt; python tools/pmc_table.py $c attn >> $OUT/r04_c2_pmc_$tag.txt); fi
done
# kernel durations of the same run (no counters)
d=/tmp/kt_enc; rm -rf $d
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $d -o p -- $P > /dev/null 2>&1
t=$(find $d -name "*kernel_trace.csv" | head -1)
(cd $ROOT && python tools/trace_by_shape.py $t 16 > $OUT/r04_c2_encoder_by_shape.txt)
# decode by shape at 64 streams x 15 s
ARGS="--chunk-s 15 --streams 64 --steps 1 --warmup 1 --no-cpu-baseline --no-pipeline-leg --no-secondary --latency-iters 0 --encoder-cus 0"
for DT in bf16 fp8a16; do
  d=/tmp/prof64_$DT; rm -rf $d
  timeout 900 rocprofv3 --kernel-trace --output-format csv -d $d -o p -- python $ROOT/bench.py --dtype $DT $ARGS > $OUT/r04_c2_b64_${DT}.json 2>/dev/null
  t=$(find $d -name "*kernel_trace.csv" | head -1)
  (cd $ROOT && python tools/trace_by_shape.py $t 14 > $OUT/r04_c2_b64_${DT}_by_shape.txt)
done
cd $ROOT
# hub leg (value_api) with prefetch A/B and the short-token regime
timeout 1200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --latency-iters 20 > $OUT/r04_c2_bench_hub.json 2> $OUT/r04_c2_bench_hub.err
python - <<PY
import json
d = json.load(open("$OUT/r04_c2_bench_hub.json"))
p = d.get("pipeline") or {}
print("value", d["value"], "step", d["roofline"]["avg_step_ms"], "value_api", d.get("value_api"))
print({k: v for k, v in p.items() if k.startswith("hub_") and k != "note"})
PY
