#!/bin/bash
# r03 GPU session P: integer-scale bilinear NCHW kernels (parity), bench, conv wgrad slices on cfg4
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_gpu_ops.py -m gpu -q -rf -k "bilinear" > gpurun_out/r03_pytest_p_ops.log 2>&1; tail -5 gpurun_out/r03_pytest_p_ops.log
timeout 600 python -m pytest tests/test_gpu_model.py tests/test_gpu_train.py -m gpu -q -rf > gpurun_out/r03_pytest_p_train.log 2>&1; tail -5 gpurun_out/r03_pytest_p_train.log
B="--no-cpu-baseline --no-roofline --no-parity --no-parity-mode --no-ref-batch --no-torch-baseline"
for i in 1 2; do
timeout 300 python bench.py --steps 6 --warmup 2 $B > gpurun_out/r03_bench_p_bf16_$i.log 2>&1
python - "$i" <<'PY'
import json, sys
for l in open(f'gpurun_out/r03_bench_p_bf16_{sys.argv[1]}.log'):
    if l.startswith('{"metric"'):
        d=json.loads(l); print('VALUE', d['value'], d['ms_per_step'], d['fwd_ms_per_img'], d['host'])
PY
done
timeout 400 python bench.py --config cfg4 --steps 4 --warmup 2 $B > gpurun_out/r03_bench_p_cfg4.log 2>&1
python - <<'PY'
import json
for l in open('gpurun_out/r03_bench_p_cfg4.log'):
    if l.startswith('{"metric"'):
        d=json.loads(l); print('cfg4 VALUE', d['value'], d['ms_per_step'], d['fwd_ms_per_img'])
PY
