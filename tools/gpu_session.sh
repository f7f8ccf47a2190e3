#!/bin/bash
# GPU session of the moment (overwritten per session; history in git).  Run as: gpurun --timeout N -- bash tools/gpu_session.sh
cd "$GRAFT_REPO_ROOT" || exit 1
REPO="$GRAFT_REPO_ROOT"; O=$REPO/gpurun_out; mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_ops.py -x -q -k "chanlogit or ln_fwd" > $O/r05_pytest_p_ops.log 2>&1; tail -3 $O/r05_pytest_p_ops.log
timeout 900 python -m pytest tests/test_gpu_configs.py tests/test_gpu_model.py tests/test_gpu_train.py tests/test_gpu_fullsize.py -x -q -k "(x3f or ns6 or gradients or bitwise) and not swin and not invpt and not cfg4 and not cfg1 and not cfg5 and not trajectory" > $O/r05_pytest_p_models.log 2>&1; tail -3 $O/r05_pytest_p_models.log
timeout 400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-torch-baseline --no-ref-batch --no-x3-mode --no-fast-mode --no-roofline > $O/r05_bench_p_chan.log 2> $O/r05_bench_p_chan.err; echo "bench rc $?"
python - <<'PY'
import json
l=[x for x in open('gpurun_out/r05_bench_p_chan.log') if x.startswith('{')][-1]
d=json.loads(l)
print({k:d[k] for k in ('value','ms_per_step','fwd_ms_per_img','peak_hbm_gb')}, d['parity']['worst_head_rel_err'])
PY
