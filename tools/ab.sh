#!/bin/bash
# One parametrised A/B runner for the GPU box (replaces the per-call scripts of rounds 3-4, tools/dbg/r04_calls/ - see tools/README.md).
#
#   bash tools/ab.sh [-o OUTFILE] [-n REPEATS] [-k KEYS] 'SETTING ...' -- COMMAND ...
#
# SETTING = comma-separated environment assignments ("-" = none), e.g.  'TW_SK_CG_MODE=0 TW_SK_CG_MODE=2,TW_SK_RING_BLOCKS=160 -'.
# The command runs once per setting (REPEATS times, interleaved: A B A B, so that clock / box drift hits both arms alike); of its
# output the LAST line that parses as JSON is reduced to KEYS (comma-separated, dotted paths: value,roofline.avg_step_ms,...),
# otherwise the last 3 lines are echoed.  Everything is appended to OUTFILE (default gpurun_out/ab.txt).
#
#   bash tools/ab.sh -o gpurun_out/r05_fuse_embed.txt -k value,roofline.avg_step_ms 'TW_FUSE_EMBED=0 TW_FUSE_EMBED=1' -- \
#        python bench.py --steps 5 --no-cpu-baseline --no-pipeline-leg --no-secondary --latency-iters 0
OUT=gpurun_out/ab.txt; REP=1; KEYS=value,ms_per_step,roofline.avg_step_ms
while getopts "o:n:k:" f; do case $f in o) OUT=$OPTARG;; n) REP=$OPTARG;; k) KEYS=$OPTARG;; *) exit 2;; esac; done
shift $((OPTIND - 1))
SETTINGS=$1; shift
[ "$1" == "--" ] && shift
mkdir -p "$(dirname "$OUT")"
echo "# $(date -u +%FT%TZ) ab.sh: $* | settings: $SETTINGS | repeats: $REP" >> "$OUT"
for r in $(seq 1 "$REP"); do
  for S in $SETTINGS; do
    ENVV=(); [ "$S" != "-" ] && IFS=, read -r -a ENVV <<< "$S"
    RES=$(env "${ENVV[@]}" "$@" 2>/tmp/ab_err.txt)
    echo "$RES" | python3 -c "
import json, sys
keys = '$KEYS'.split(',')
lines = [l for l in sys.stdin.read().splitlines() if l.strip()]
d = None
for l in reversed(lines):
    try:
        d = json.loads(l); break
    except Exception:
        pass
if isinstance(d, dict):
    def get(o, path):
        for p in path.split('.'):
            o = o.get(p) if isinstance(o, dict) else None
        return o
    print('[$S] run $r:', ', '.join(f'{k}={get(d, k)}' for k in keys))
else:
    print('[$S] run $r:', ' | '.join(lines[-3:]))
" | tee -a "$OUT"
  done
done
