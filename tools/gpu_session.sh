#!/bin/bash
# GPU session of the moment (overwritten per session; history in git).  Run as: gpurun --timeout N -- bash tools/gpu_session.sh
cd "$GRAFT_REPO_ROOT" || exit 1
REPO="$GRAFT_REPO_ROOT"; O=$REPO/gpurun_out; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -k "bn_ or upconv" > $O/r05_pytest_c_ops.log 2>&1; echo "pytest rc $?" >> $O/r05_pytest_c_ops.log
tail -4 $O/r05_pytest_c_ops.log
timeout 900 python -m pytest tests/test_gpu_train.py tests/test_gpu_fullsize.py -x -q -k "not swin and not invpt and not cfg4 and not trajectory" > $O/r05_pytest_c_train.log 2>&1; echo "pytest rc $?" >> $O/r05_pytest_c_train.log
tail -6 $O/r05_pytest_c_train.log
timeout 400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-torch-baseline --no-ref-batch --no-x3-mode --no-fast-mode --no-parity > $O/r05_bench_c_headnode.log 2> $O/r05_bench_c_headnode.err; echo "bench rc $?"
python - <<'PY'
import json
l=[x for x in open('gpurun_out/r05_bench_c_headnode.log') if x.startswith('{')][-1]
d=json.loads(l)
print({k:d[k] for k in ('value','ms_per_step','fwd_ms_per_img','peak_hbm_gb')})
PY
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_c -o c -- python $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-torch-baseline --no-ref-batch --no-x3-mode --no-fast-mode --no-parity --no-fwd --no-roofline > $O/r05_prof_c_run.log 2>&1
python $REPO/tools/prof_summary.py /tmp/prof_c 5 > $O/r05_train_ns6_b63_x3f_c.txt 2>&1
head -50 $O/r05_train_ns6_b63_x3f_c.txt
