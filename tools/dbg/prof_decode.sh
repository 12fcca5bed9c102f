cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_dec
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_dec -o dec -- python $GRAFT_REPO_ROOT/tools/bench_decode.py --batches ${1:-16} > /tmp/prof_dec.log 2>&1
f=$(find /tmp/prof_dec -name "*kernel_stats.csv" | head -1)
cd $GRAFT_REPO_ROOT && python tools/summarize_rocprof.py $f | head -16
grep "^B=" /tmp/prof_dec.log
