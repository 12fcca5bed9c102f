#!/bin/bash
# round 4, call 29 (last): do the two workgroups of a CU get out of each other's way when one of them starts late?  Scratch builds of
# k_gemm.hip in which the workgroups 256..511 (map 0) / every other workgroup of an XCD among the first 512 (map 1) sleep 2 or 4 x
# s_sleep(127) (3.4 us each) before their first tile; product library as the baseline
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
for L in "" libtw_dephase_2_0.so libtw_dephase_4_0.so libtw_dephase_2_1.so libtw_dephase_4_1.so ""; do
  echo "== ${L:-product}"
  if [ -n "$L" ]; then export THEWHISPER_LIB=$ROOT/thewhisper_amd/lib/dbg/$L; else unset THEWHISPER_LIB; fi
  timeout 100 python tools/bench_encoder.py --cases 500x16,1500x16 2>&1 | grep encode_ms | cut -c1-70
done | tee $OUT/r04_c29_dephase.txt
