#!/bin/bash
# r03 GPU: last sanity after the eval-head cast change: model tests, smoke, NS-6 config parity, short bench
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_model.py tests/test_gpu_configs.py -m gpu -q -rf -k "not cfg5 and not swin" > gpurun_out/r03_pytest_fin3.log 2>&1; tail -3 gpurun_out/r03_pytest_fin3.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
B="--no-cpu-baseline --no-roofline --no-parity --no-parity-mode --no-ref-batch --no-torch-baseline"
timeout 300 python bench.py --steps 6 --warmup 2 $B > gpurun_out/r03_bench_fin3.log 2>&1
python - <<'PY'
import json
for l in open('gpurun_out/r03_bench_fin3.log'):
    if l.startswith('{"metric"'):
        d=json.loads(l); print('VALUE', d['value'], d['ms_per_step'], d['fwd_ms_per_img'])
PY
