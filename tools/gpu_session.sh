#!/bin/bash
# round 6, session 41: MlpHalfFn on split planes for channel counts % 32 (InvPT's 288-channel stage): cfg4 parity + same-box A/B
cd "$GRAFT_REPO_ROOT" || exit 1
REPO="$GRAFT_REPO_ROOT"; O=$REPO/gpurun_out; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -k "cfg4 or invpt or cfg1" 2>&1 | tail -3
B="--no-torch-baseline --no-cpu-baseline --no-ref-batch --no-x3-mode --no-fast-mode --no-parity --no-roofline"
for rep in 1 2; do
for v in 0 1; do
  MTT_MLP_SPLIT_RULE64=$v timeout 900 python bench.py --config cfg4 --steps 5 --warmup 2 $B > $O/r06_bench_ar_cfg4_rule64_$v.log 2>$O/r06_bench_ar_cfg4_rule64_$v.err
  python - $O/r06_bench_ar_cfg4_rule64_$v.log "cfg4 rule64=$v" <<'PY'
import json, sys
l=[x for x in open(sys.argv[1]) if x.startswith('{')]
if l:
    d=json.loads(l[-1]); print(sys.argv[2], d['config']['per_gpu_batch'], {k:d.get(k) for k in ('value','ms_per_step','fwd_ms_per_img','peak_hbm_gb')})
else: print(sys.argv[2], 'NO LINE', open(sys.argv[1].replace('.log','.err')).read()[-900:])
PY
done
done
