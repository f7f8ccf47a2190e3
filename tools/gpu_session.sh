#!/bin/bash
# round 6, session 39: x3 window-attention forward with K split once per workgroup into LDS: parity + kernel time + Swin bench
cd "$GRAFT_REPO_ROOT" || exit 1
REPO="$GRAFT_REPO_ROOT"; O=$REPO/gpurun_out; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -q -k "winattn" 2>&1 | tail -2; echo ops-done
timeout 1500 python -m pytest tests -m gpu -q -k "swin or Swin" 2>&1 | tail -2
B="--no-torch-baseline --no-cpu-baseline --no-ref-batch --no-x3-mode --no-fast-mode --no-parity --no-roofline"
timeout 900 python bench.py --config swinb --steps 5 --warmup 2 $B > $O/r06_bench_ap_swinb.log 2>$O/r06_bench_ap_swinb.err
python - $O/r06_bench_ap_swinb.log <<'PY'
import json, sys
l=[x for x in open(sys.argv[1]) if x.startswith('{')]
if l:
    d=json.loads(l[-1]); print('swinb', d['config']['per_gpu_batch'], {k:d.get(k) for k in ('value','ms_per_step','fwd_ms_per_img','peak_hbm_gb')})
PY
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof_sw -o swin -- python $REPO/bench.py --config swinb --steps 2 --warmup 1 --no-fwd $B > $O/r06_prof_ap_swin_run.log 2>&1
cd $REPO
python tools/prof_summary.py /tmp/prof_sw 3 > $O/r06_train_swinb_b16_x3f_ap.txt 2>/dev/null; grep -i "winattn" $O/r06_train_swinb_b16_x3f_ap.txt | cut -c1-150; head -3 $O/r06_train_swinb_b16_x3f_ap.txt | cut -c1-120
