#!/bin/bash
# GPU session of the moment (overwritten per session; history in git).  Run as: gpurun --timeout N -- bash tools/gpu_session.sh
cd "$GRAFT_REPO_ROOT" || exit 1
REPO="$GRAFT_REPO_ROOT"; O=$REPO/gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_train.py tests/test_gpu_fullsize.py tests/test_gpu_model.py -x -q -k "invpt or cfg4" > $O/r05_pytest_f_invpt.log 2>&1; echo "pytest rc $?" >> $O/r05_pytest_f_invpt.log
tail -5 $O/r05_pytest_f_invpt.log
timeout 420 python bench.py --config cfg4 --steps 8 --warmup 2 --no-cpu-baseline --no-torch-baseline --no-ref-batch --no-x3-mode --no-fast-mode --no-parity > $O/r05_bench_f_cfg4.log 2> $O/r05_bench_f_cfg4.err; echo "cfg4 rc $?"
python - <<'PY'
import json
l = [x for x in open('gpurun_out/r05_bench_f_cfg4.log') if x.startswith('{')][-1]
d = json.loads(l)
print('cfg4 x3f', d['value'], 'img/s', d['ms_per_step'], 'ms fwd', d['fwd_ms_per_img'], d['peak_hbm_gb'])
PY
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_f -o f -- python $REPO/bench.py --config cfg4 --steps 3 --warmup 1 --no-cpu-baseline --no-torch-baseline --no-ref-batch --no-x3-mode --no-fast-mode --no-parity --no-fwd --no-roofline > $O/r05_prof_f_run.log 2>&1
python $REPO/tools/prof_summary.py /tmp/prof_f 5 > $O/r05_train_cfg4_b32_x3f_f.txt 2>&1
head -45 $O/r05_train_cfg4_b32_x3f_f.txt
