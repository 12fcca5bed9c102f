cd $GRAFT_REPO_ROOT
run() { # label, env...
  lbl=$1; shift
  for cfg in "large-v3-turbo 30 1" "large-v3 10 16" "large-v3 30 16"; do
    set -- $cfg "$@"
    m=$1; c=$2; s=$3; shift 3
    env "$@" python bench.py --model $m --chunk-s $c --streams $s --steps 6 --warmup 2 --no-cpu-baseline --no-secondary --no-pipeline-leg --latency-iters 0 --encoder-cus 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); st=d['stage_ms_per_step']
print('$lbl', '$m', $c, $s, 'enc', st['encode_ms'], 'ckv', st['cross_kv_ms'], 'tok/s', d['value'])"
  done
}
run base X=1
run cfg8 TW_GEMM_CFG=8
run cfg5 TW_GEMM_CFG=5
run cfg6 TW_GEMM_CFG=6
run cfg1 TW_GEMM_CFG=1
run narrow8 TW_GEMM_NARROW=8
