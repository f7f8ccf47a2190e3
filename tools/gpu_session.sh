#!/bin/bash
# round 6, session 18: integer-scale NHWC bilinear backward (cfg4's InvPT stage resizes): op parity, model parity, cfg4 bench + step profile
cd "$GRAFT_REPO_ROOT" || exit 1
REPO="$GRAFT_REPO_ROOT"; O=$REPO/gpurun_out; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "bilinear" 2>&1 | tail -3
timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_train.py -m gpu -x -q 2>&1 | tail -3
timeout 1200 python bench.py --config cfg4 --steps 8 --warmup 2 --no-torch-baseline --no-cpu-baseline --no-ref-batch --no-x3-mode > $O/r06_bench_s_cfg4.log 2> $O/r06_bench_s_cfg4.err; echo "cfg4 rc $?"
python - <<'PY'
import json
l=[x for x in open('gpurun_out/r06_bench_s_cfg4.log') if x.startswith('{')]
d=json.loads(l[-1]); f=d.get('fast_mode') or {}
print('cfg4', {k:d[k] for k in ('value','ms_per_step','fwd_ms_per_img','peak_hbm_gb')}, 'bf16', f.get('images_per_s'), 'parity', (d.get('parity') or {}).get('worst_head_rel_err'))
PY
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof_s -o cfg4 -- python $REPO/bench.py --config cfg4 --steps 3 --warmup 1 --no-torch-baseline --no-cpu-baseline --no-ref-batch --no-x3-mode --no-fast-mode --no-parity --no-roofline > $O/r06_prof_s_run.log 2>&1; echo "prof rc $?"
cd $REPO
python tools/prof_summary.py /tmp/prof_s 3 > $O/r06_train_cfg4_b32_x3f_s.txt 2>&1 || ls -R /tmp/prof_s | head
head -30 $O/r06_train_cfg4_b32_x3f_s.txt
