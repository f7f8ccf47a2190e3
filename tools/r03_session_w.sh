#!/bin/bash
# r03 GPU session W: ctr_mix bf16 output (parity), training tests, bf16 bench
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_gpu_ops.py -m gpu -q -rf -k "ctr_" > gpurun_out/r03_pytest_w_ops.log 2>&1; tail -3 gpurun_out/r03_pytest_w_ops.log
timeout 900 python -m pytest tests/test_gpu_train.py tests/test_gpu_model.py -m gpu -q -rf > gpurun_out/r03_pytest_w.log 2>&1; tail -3 gpurun_out/r03_pytest_w.log
B="--no-cpu-baseline --no-roofline --no-parity --no-parity-mode --no-ref-batch --no-torch-baseline"
timeout 300 python bench.py --steps 6 --warmup 2 $B > gpurun_out/r03_bench_w_bf16.log 2>&1
python - <<'PY'
import json
for l in open('gpurun_out/r03_bench_w_bf16.log'):
    if l.startswith('{"metric"'):
        d=json.loads(l); print('VALUE', d['value'], d['ms_per_step'], d['fwd_ms_per_img'], d['host'])
PY
