#!/bin/bash
# r03 GPU session V: x3f decoder backward on bf16 copies: x3f tests, x3f bench line
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_train.py tests/test_gpu_fullsize.py -m gpu -q -rf -k "x3f or reproducible" > gpurun_out/r03_pytest_v.log 2>&1; tail -5 gpurun_out/r03_pytest_v.log
B="--no-cpu-baseline --no-roofline --no-parity --no-parity-mode --no-ref-batch --no-torch-baseline"
timeout 300 python bench.py --prec x3f --steps 4 --warmup 2 $B > gpurun_out/r03_bench_v_x3f.log 2>&1
python - <<'PY'
import json
for l in open('gpurun_out/r03_bench_v_x3f.log'):
    if l.startswith('{"metric"'):
        d=json.loads(l); print('x3f VALUE', d['value'], d['ms_per_step'], d['fwd_ms_per_img'], d['peak_hbm_gb'])
PY
