cd $GRAFT_REPO_ROOT
for e in TW_GEMM_WREG_MIN=384 TW_GEMM_WREG_MIN=512 TW_GEMM_WREG_MIN=384 TW_GEMM_WREG_MIN=512 TW_GEMM_WREG_MIN=700; do
  for cfg in "large-v3-turbo 30 1" "large-v3 10 4" "large-v3 10 3" "large-v3 15 2" "large-v3 10 16"; do
    set -- $cfg; m=$1; c=$2; s=$3
    env $e python bench.py --model $m --chunk-s $c --streams $s --steps 10 --warmup 2 --no-cpu-baseline --no-secondary --no-pipeline-leg --latency-iters 0 --encoder-cus 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); st=d['stage_ms_per_step']
print('$e', '$m', $c, $s, 'enc', st['encode_ms'], 'ckv', st['cross_kv_ms'], 'tok/s', d['value'])"
  done
done
