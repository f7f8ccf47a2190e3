#!/bin/bash
# round 6, session 34: Swin chan_kv as a split-K autograd node: parity, same-box A/B
cd "$GRAFT_REPO_ROOT" || exit 1
REPO="$GRAFT_REPO_ROOT"; O=$REPO/gpurun_out; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -k "swin or Swin" 2>&1 | tail -4
B="--no-torch-baseline --no-cpu-baseline --no-ref-batch --no-x3-mode --no-fast-mode --no-parity --no-roofline"
show() { python - $1 "$2" <<'PY'
import json, sys
l=[x for x in open(sys.argv[1]) if x.startswith('{')]
if l:
    d=json.loads(l[-1]); print(sys.argv[2], d['config']['per_gpu_batch'], {k:d.get(k) for k in ('value','ms_per_step','fwd_ms_per_img','peak_hbm_gb')})
else: print(sys.argv[2], 'NO LINE', open(sys.argv[1].replace('.log','.err')).read()[-900:])
PY
}
for rep in 1 2; do
for v in 1 0; do
  MTT_CHAN_KV_FN=$v timeout 900 python bench.py --config swinb --steps 5 --warmup 2 $B > $O/r06_bench_ak_swinb_ckv$v.log 2>$O/r06_bench_ak_swinb_ckv$v.err; show $O/r06_bench_ak_swinb_ckv$v.log "swinb x3f chan_kv_fn=$v"
done
done
for v in 1 0; do
  MTT_CHAN_KV_FN=$v timeout 900 python bench.py --config swinb --prec bf16 --steps 5 --warmup 2 $B > $O/r06_bench_ak_swinb_bf16_ckv$v.log 2>$O/r06_bench_ak_swinb_bf16_ckv$v.err; show $O/r06_bench_ak_swinb_bf16_ckv$v.log "swinb bf16 chan_kv_fn=$v"
done
