#!/bin/bash
# Round profile on the GPU box (run through gpurun from the repo root):  bash tools/profile_round.sh r02 [quick]
# Produces in gpurun_out/: bench lines (default with every BASELINE config as a leg, turbo 30 s B=1 with overlap), rocprofv3 kernel stats + per-shape split
# of the default bench, PMC passes (FETCH_SIZE, WRITE_SIZE -> HBM bytes per decode step; MFMA busy of the encoder kernels;
# each in its own --pmc + --kernel-trace pass).  Raw traces stay in /tmp.
set -u
R=${1:-r02}
QUICK=${2:-}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# the default line carries every BASELINE configuration since round 3 (other_configs: turbo 30 s x 1, fp8 / bf16 15 s x 16; rtfx)
python $ROOT/bench.py --steps 20 > $OUT/${R}_bench_default.json 2> $OUT/${R}_bench_default.err
if [ -z "$QUICK" ]; then
python $ROOT/bench.py --model large-v3-turbo --chunk-s 30 --streams 1 --no-cpu-baseline --no-secondary --latency-iters 30 > $OUT/${R}_bench_turbo_30s_b1.json 2>/dev/null
fi
rm -rf /tmp/prof_stats
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -o p -- python $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-pipeline-leg --no-secondary --latency-iters 0 > $OUT/${R}_bench_profiled_run.json 2>/dev/null
f=$(find /tmp/prof_stats -name "*kernel_stats.csv" | head -1)
t=$(find /tmp/prof_stats -name "*kernel_trace.csv" | head -1)
cp $f $OUT/${R}_large-v3_b16_kernel_stats.csv
(cd $ROOT && python tools/summarize_rocprof.py $f > $OUT/${R}_kernel_table.md; python tools/trace_by_shape.py $t 30 > $OUT/${R}_large-v3_b16_by_shape.txt)
# the same WITHOUT the encoder overlap (one context, stages back to back): the per-step kernel sum of THIS trace is what the HIP-event
# step time of a decode-only run is compared with (under the overlap the encoder's launches on the other CUs stretch the decode kernels)
rm -rf /tmp/prof_dec
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_dec -o p -- python $ROOT/bench.py --steps 2 --warmup 1 --encoder-cus 0 --no-cpu-baseline --no-pipeline-leg --no-secondary --latency-iters 0 > $OUT/${R}_bench_profiled_run_decode_only.json 2>/dev/null
t=$(find /tmp/prof_dec -name "*kernel_trace.csv" | head -1)
(cd $ROOT && python tools/trace_by_shape.py $t 30 > $OUT/${R}_large-v3_b16_by_shape_decode_only.txt)
PMCARGS="--steps 1 --warmup 0 --no-graph --encoder-cus 0 --no-cpu-baseline --no-pipeline-leg --no-secondary --latency-iters 0"
for C in FETCH_SIZE WRITE_SIZE; do
  d=/tmp/prof_$C; rm -rf $d
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $d -o p -- python $ROOT/bench.py $PMCARGS > /dev/null 2>&1
  c=$(find $d -name "*counter_collection.csv" | head -1)
  (cd $ROOT && python tools/pmc_summary.py $c $C > $OUT/${R}_large-v3_b16_pmc_$C.txt)
done
(cd $ROOT && python tools/pmc_step_traffic.py $(find /tmp/prof_FETCH_SIZE -name "*counter_collection.csv" | head -1) \
   $(find /tmp/prof_WRITE_SIZE -name "*counter_collection.csv" | head -1) \
   '{"model": "large-v3", "streams": 16, "chunk_s": 10, "dtype": "f16", "new_tokens": 128}' > $OUT/${R}_pmc_step_traffic.json)   # (bench.py's default dtype since round 6)
d=/tmp/prof_mfma; rm -rf $d
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $d -o p -- python $ROOT/bench.py $PMCARGS --new-tokens 8 > /dev/null 2> $OUT/${R}_pmc_mfma.err
c=$(find $d -name "*counter_collection.csv" | head -1)
(cd $ROOT && python tools/pmc_mfma_busy.py $c > $OUT/${R}_large-v3_b16_pmc_mfma_busy.txt 2>> $OUT/${R}_pmc_mfma.err)
ls -la $OUT | tail -20
# the floors the projection launches are compared with (same box): a launch that only streams the same bytes
if [ -x $ROOT/tools/dbg/stream_floor ]; then $ROOT/tools/dbg/stream_floor > $OUT/${R}_stream_floor.txt 2>&1; fi
