#!/bin/bash
# round 5, call 3: two tiles per workgroup (DUAL ring) for the 320-tile launches, 16-row tiles for the narrow launches above 16 streams,
# the embedding fused into the sampler; per-kernel split of a 64- and a 32-stream decode step
set -x
O=gpurun_out/r05_call3; mkdir -p $O
cd tools/dbg
for m in 2 3; do PROBE_CUS=160 TW_SK_CG_MODE=$m ./probe_gemv_new > ../../$O/probe_mode$m.txt 2>&1; done
for m in 2 3; do PROBE_CUS=160 PROBE_TR_NARROW=16 TW_SK_CG_MODE=$m ./probe_gemv_new > ../../$O/probe_mode${m}_narrow16.txt 2>&1; done
PROBE_CUS=160 PROBE_B=64 TW_SK_CG_MODE=3 ./probe_gemv_new_ts > ../../$O/probe_ts_b64_mode3.txt 2>&1
PROBE_CUS=160 PROBE_B=64 TW_SK_CG_MODE=2 ./probe_gemv_new_ts > ../../$O/probe_ts_b64_mode2.txt 2>&1
PROBE_CUS=160 PROBE_B=16 ./probe_gemv_new_ts > ../../$O/probe_ts_b16.txt 2>&1
cd ../..
grep "sum of" $O/probe_mode*.txt
timeout 900 python -m pytest tests -m gpu -x -q -s > $O/gpu_tests.log 2>&1; echo "pytest rc=$?" >> $O/gpu_tests.log
tail -5 $O/gpu_tests.log
for m in 3 2; do TW_SK_CG_MODE=$m timeout 300 python tools/bench_decode.py --layers 32 --batches 16,32,64 --tokens 128 > $O/decode_mode$m.txt 2>&1; tail -3 $O/decode_mode$m.txt; done
TW_FUSE_EMBED=0 timeout 300 python tools/bench_decode.py --layers 32 --batches 1,16 --tokens 128 > $O/decode_nofuse_embed.txt 2>&1; tail -2 $O/decode_nofuse_embed.txt
timeout 300 python tools/bench_decode.py --layers 32 --batches 1,16 --tokens 128 > $O/decode_fuse_embed.txt 2>&1; tail -2 $O/decode_fuse_embed.txt
cd /tmp && export TMPDIR=/tmp
for B in 64 32; do
  rm -rf /tmp/prof_b$B
  rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_b$B -o p -- python $GRAFT_REPO_ROOT/tools/bench_decode.py --layers 32 --batches $B --tokens 128 > /dev/null 2>&1
  t=$(find /tmp/prof_b$B -name "*kernel_trace.csv" | head -1)
  (cd $GRAFT_REPO_ROOT && python tools/trace_by_shape.py $t 40 > $O/b${B}_10s_by_shape.txt)
done
head -30 $GRAFT_REPO_ROOT/$O/b64_10s_by_shape.txt
