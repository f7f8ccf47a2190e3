#!/bin/bash
# r03 GPU session Z: modulate_bwd write semantics, vectorised row casts: parity, training tests, bf16 + x3f bench lines
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_gpu_ops.py -m gpu -q -rf -k "modulate or rowscale or cast" > gpurun_out/r03_pytest_z_ops.log 2>&1; tail -3 gpurun_out/r03_pytest_z_ops.log
timeout 900 python -m pytest tests/test_gpu_train.py tests/test_gpu_model.py tests/test_gpu_fullsize.py -m gpu -q -rf -k "not cfg5" > gpurun_out/r03_pytest_z.log 2>&1; tail -3 gpurun_out/r03_pytest_z.log
B="--no-cpu-baseline --no-roofline --no-parity --no-parity-mode --no-ref-batch --no-torch-baseline"
timeout 300 python bench.py --steps 6 --warmup 2 $B > gpurun_out/r03_bench_z_bf16.log 2>&1
timeout 300 python bench.py --prec x3f --steps 4 --warmup 2 $B > gpurun_out/r03_bench_z_x3f.log 2>&1
python - <<'PY'
import json
for f in ('bf16','x3f'):
    for l in open(f'gpurun_out/r03_bench_z_{f}.log'):
        if l.startswith('{"metric"'):
            d=json.loads(l); print(f, 'VALUE', d['value'], d['ms_per_step'], d['fwd_ms_per_img'])
PY
