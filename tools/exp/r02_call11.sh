#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
( timeout 1500 python -m pytest tests -q -m gpu -s 2>&1 | grep -E "FULLDEPTH|passed|failed|rror|max word-timestamp" | cut -c1-900 ) > $OUT/c11_allgpu.log
( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 ) > $OUT/c11_smoke.log
bash tools/profile_round.sh r02 > $OUT/c11_profile.log 2>&1
( timeout 900 python benchmark/run_rtfx.py 2>/dev/null | tail -1 ) > $OUT/r02_rtfx_turbo.json
cat $OUT/c11_allgpu.log | tail -12; cat $OUT/c11_smoke.log; tail -25 $OUT/c11_profile.log
