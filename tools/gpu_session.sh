#!/bin/bash
# round 6, session 36: window-attention backward with the transposed bias table in the key-owner pass (ABI 13): parity, same-box A/B
cd "$GRAFT_REPO_ROOT" || exit 1
REPO="$GRAFT_REPO_ROOT"; O=$REPO/gpurun_out; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -q -k "winattn" 2>&1 | tail -3
timeout 1500 python -m pytest tests -m gpu -q -k "swin or Swin" 2>&1 | tail -3
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
  MTT_WINATTN_BIAST=$v timeout 900 python bench.py --config swinb --steps 5 --warmup 2 $B > $O/r06_bench_am_swinb_bt$v.log 2>$O/r06_bench_am_swinb_bt$v.err; show $O/r06_bench_am_swinb_bt$v.log "swinb x3f biasT=$v"
done
done
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof_sw -o swin -- python $REPO/bench.py --config swinb --steps 2 --warmup 1 --no-fwd $B > $O/r06_prof_am_swin_run.log 2>&1
cd $REPO
python tools/prof_summary.py /tmp/prof_sw 3 2>/dev/null | grep -i "winattn" | cut -c1-150
