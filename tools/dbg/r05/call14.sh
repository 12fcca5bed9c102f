#!/bin/bash
# round 5, call 14: HBM traffic and L2 hit rate of a 64-stream decode step (does the activation block every workgroup re-reads come from
# L2 or from HBM?) - separate --pmc passes with --kernel-trace only, as the microarchitecture guide prescribes
O=$GRAFT_REPO_ROOT/gpurun_out/r05_call14; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
ARGS="--streams 64 --steps 1 --warmup 0 --no-graph --encoder-cus 0 --no-cpu-baseline --no-pipeline-leg --no-secondary --latency-iters 0 --new-tokens 24"
for C in FETCH_SIZE WRITE_SIZE; do
  d=/tmp/prof64_$C; rm -rf $d
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $d -o p -- python $GRAFT_REPO_ROOT/bench.py $ARGS > /dev/null 2>&1
done
(cd $GRAFT_REPO_ROOT && python tools/pmc_step_traffic.py $(find /tmp/prof64_FETCH_SIZE -name "*counter_collection.csv" | head -1) $(find /tmp/prof64_WRITE_SIZE -name "*counter_collection.csv" | head -1) \
   '{"model": "large-v3", "streams": 64, "chunk_s": 10, "dtype": "bf16", "new_tokens": 24}' > $O/b64_pmc_step_traffic.json)
d=/tmp/prof64_tcc; rm -rf $d
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --kernel-trace --output-format csv -d $d -o p -- python $GRAFT_REPO_ROOT/bench.py $ARGS > /dev/null 2> $O/tcc.err
(cd $GRAFT_REPO_ROOT && python tools/pmc_table.py $(find $d -name "*counter_collection.csv" | head -1) skinny > $O/b64_pmc_tcc_skinny.txt; python tools/pmc_table.py $(find $d -name "*counter_collection.csv" | head -1) dec_ >> $O/b64_pmc_tcc_skinny.txt)
head -c 1500 $O/b64_pmc_step_traffic.json; head -12 $O/b64_pmc_tcc_skinny.txt | cut -c1-200
