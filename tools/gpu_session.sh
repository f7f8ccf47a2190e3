#!/bin/bash
# round 6, session 25: Swin-B after the glue fixes (views instead of selects in the heads, one residual node per block, one multi-scale sum node)
cd "$GRAFT_REPO_ROOT" || exit 1
REPO="$GRAFT_REPO_ROOT"; O=$REPO/gpurun_out; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -k "swin or Swin or deconv or cfg5" 2>&1 | tail -6
B="--no-torch-baseline --no-cpu-baseline --no-ref-batch --no-x3-mode --no-fast-mode --no-parity --no-roofline"
for pr in x3f bf16; do
  timeout 900 python bench.py --config swinb --steps 4 --warmup 1 --prec $pr $B > $O/r06_bench_z_swinb_$pr.log 2> $O/r06_bench_z_swinb_$pr.err; echo "swinb $pr rc $?"
  python - $O/r06_bench_z_swinb_$pr.log <<'PY'
import json, sys
l=[x for x in open(sys.argv[1]) if x.startswith('{')]
if l:
    d=json.loads(l[-1]); print({k:d.get(k) for k in ('value','ms_per_step','fwd_ms_per_img','peak_hbm_gb')})
else: print(open(sys.argv[1].replace('.log','.err')).read()[-1500:])
PY
done
timeout 900 python bench.py --config cfg5 --steps 4 --warmup 1 $B > $O/r06_bench_z_cfg5.log 2> $O/r06_bench_z_cfg5.err; echo "cfg5 rc $?"
python - $O/r06_bench_z_cfg5.log <<'PY'
import json, sys
l=[x for x in open(sys.argv[1]) if x.startswith('{')]
if l:
    d=json.loads(l[-1]); print({k:d.get(k) for k in ('value','ms_per_step','fwd_ms_per_img','peak_hbm_gb')})
PY
timeout 900 python tools/torch_ops_profile.py 8 x3f swinb > $O/r06_torch_ops_z_swinb.log 2>&1; echo rc $?
grep -A 45 "aten ops with device" $O/r06_torch_ops_z_swinb.log | cut -c1-200
