#!/bin/bash
# r03 GPU session J: the 128 x 128 LDS-DMA kernel (parity, decoder shapes vs the other kernels), chanlogit px8, bench
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -q -rf -k "gemm or chanlogit" > gpurun_out/r03_pytest_j_ops.log 2>&1; tail -6 gpurun_out/r03_pytest_i_ops.log
timeout 300 python tools/dec_gemm_bench.py > gpurun_out/r03_dec_gemm_bench_j.log 2>&1; tail -30 gpurun_out/r03_dec_gemm_bench_j.log
B="--no-cpu-baseline --no-roofline --no-parity --no-parity-mode --no-ref-batch --no-torch-baseline"
timeout 300 python bench.py --steps 6 --warmup 2 $B > gpurun_out/r03_bench_j_bf16.log 2>&1; tail -c 900 gpurun_out/r03_bench_j_bf16.log | head -c 400; echo
python - <<'PY'
import json
for l in open('gpurun_out/r03_bench_j_bf16.log'):
    if l.startswith('{"metric"'):
        d=json.loads(l); print('VALUE', d['value'], d['ms_per_step'], d['fwd_ms_per_img'], d['host'])
PY
