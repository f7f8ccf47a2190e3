#!/bin/bash
# round 6, session 27: window attention backward with LDS-staged operands; cfg4 after the pitch threshold fix (160); Swin bench + profile
cd "$GRAFT_REPO_ROOT" || exit 1
REPO="$GRAFT_REPO_ROOT"; O=$REPO/gpurun_out; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -q -k "winattn" 2>&1 | tail -4
timeout 1500 python -m pytest tests -m gpu -q -k "swin or Swin or cfg4 or invpt" 2>&1 | tail -6
B="--no-torch-baseline --no-cpu-baseline --no-ref-batch --no-x3-mode"
timeout 1200 python bench.py --config cfg4 --steps 8 --warmup 2 $B > $O/r06_bench_ab_cfg4.log 2> $O/r06_bench_ab_cfg4.err; echo "cfg4 rc $?"
python - $O/r06_bench_ab_cfg4.log cfg4 <<'PY'
import json, sys
l=[x for x in open(sys.argv[1]) if x.startswith('{')]
if l:
    d=json.loads(l[-1]); f=d.get('fast_mode') or {}
    print(sys.argv[2], {k:d.get(k) for k in ('value','ms_per_step','fwd_ms_per_img','peak_hbm_gb')}, 'bf16', f.get('images_per_s'), f.get('fwd_ms_per_img'), 'parity', (d.get('parity') or {}).get('worst_head_rel_err'))
else: print(open(sys.argv[1].replace('.log','.err')).read()[-1500:])
PY
B="$B --no-fast-mode --no-parity --no-roofline"
for pr in x3f bf16; do
  timeout 900 python bench.py --config swinb --steps 4 --warmup 1 --prec $pr $B > $O/r06_bench_ab_swinb_$pr.log 2> $O/r06_bench_ab_swinb_$pr.err; echo "swinb $pr rc $?"
  python - $O/r06_bench_ab_swinb_$pr.log <<'PY'
import json, sys
l=[x for x in open(sys.argv[1]) if x.startswith('{')]
if l:
    d=json.loads(l[-1]); print({k:d.get(k) for k in ('value','ms_per_step','fwd_ms_per_img','peak_hbm_gb')})
else: print(open(sys.argv[1].replace('.log','.err')).read()[-1500:])
PY
done
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof_sw -o swin -- python $REPO/bench.py --config swinb --steps 2 --warmup 1 --no-fwd $B > $O/r06_prof_ab_swin_run.log 2>&1; echo "prof swin rc $?"
cd $REPO
python tools/prof_summary.py /tmp/prof_sw 3 > $O/r06_train_swinb_b8_x3f_ab.txt 2>&1
head -24 $O/r06_train_swinb_b8_x3f_ab.txt | cut -c1-180
