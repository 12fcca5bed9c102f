#!/bin/bash
# Round profile on the GPU box (run through gpurun from the repo root):  bash tools/profile_round.sh r01
# Produces in gpurun_out/: bench lines (default, fp8 15 s, turbo 30 s B=1), rocprofv3 kernel stats + per-shape split of the
# default bench, PMC passes (FETCH_SIZE, WRITE_SIZE; own passes, kernel-trace only).  Raw traces stay in /tmp.
set -u
R=${1:-r01}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python $ROOT/bench.py > $OUT/${R}_bench_default.json 2> $OUT/${R}_bench_default.err
python $ROOT/bench.py --dtype fp8 --chunk-s 15 --no-cpu-baseline --latency-iters 30 > $OUT/${R}_bench_fp8_15s.json 2>/dev/null
python $ROOT/bench.py --model large-v3-turbo --chunk-s 30 --streams 1 --no-cpu-baseline --latency-iters 30 > $OUT/${R}_bench_turbo_30s_b1.json 2>/dev/null
python $ROOT/bench.py --chunk-s 30 --streams 8 --no-cpu-baseline --latency-iters 0 > $OUT/${R}_bench_30s_b8.json 2>/dev/null
rm -rf /tmp/prof_stats /tmp/prof_fetch /tmp/prof_write
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -o p -- python $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --latency-iters 3 > $OUT/${R}_bench_profiled_run.json 2>/dev/null
f=$(find /tmp/prof_stats -name "*kernel_stats.csv" | head -1)
t=$(find /tmp/prof_stats -name "*kernel_trace.csv" | head -1)
cp $f $OUT/${R}_large-v3_b16_kernel_stats.csv
(cd $ROOT && python tools/summarize_rocprof.py $f > $OUT/${R}_kernel_table.md; python tools/trace_by_shape.py $t 30 > $OUT/${R}_large-v3_b16_by_shape.txt)
for C in FETCH_SIZE WRITE_SIZE; do
  d=/tmp/prof_$C; rm -rf $d
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $d -o p -- python $ROOT/bench.py --steps 1 --warmup 0 --no-graph --no-cpu-baseline --latency-iters 0 --new-tokens 16 > /dev/null 2>&1
  c=$(find $d -name "*counter_collection.csv" | head -1)
  (cd $ROOT && python tools/pmc_summary.py $c $C > $OUT/${R}_large-v3_b16_pmc_$C.txt)
done
ls -la $OUT | tail -15
