#!/bin/bash
# r03 GPU session S: small long-K GEMMs on the LDS-DMA 128 kernel (narrow tiles): bench + parity + training tests + bench line
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
true
timeout 600 python -m pytest tests/test_gpu_train.py tests/test_gpu_model.py tests/test_gpu_fullsize.py -m gpu -q -rf -x -k "not cfg4 and not cfg5 and not swin" > gpurun_out/r03_pytest_s.log 2>&1; tail -4 gpurun_out/r03_pytest_s.log
B="--no-cpu-baseline --no-roofline --no-parity --no-parity-mode --no-ref-batch --no-torch-baseline"
timeout 300 python bench.py --steps 6 --warmup 2 $B > gpurun_out/r03_bench_s_bf16.log 2>&1
python - <<'PY'
import json
for l in open('gpurun_out/r03_bench_s_bf16.log'):
    if l.startswith('{"metric"'):
        d=json.loads(l); print('VALUE', d['value'], d['ms_per_step'], d['fwd_ms_per_img'], d['host'])
PY
