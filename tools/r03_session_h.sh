#!/bin/bash
# r03 GPU session H2: decoder forward / dgrad GEMMs general vs LDS-DMA kernel; decoder wgrads now on the token-major kernel (parity tests)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 300 python tools/dec_gemm_bench.py > gpurun_out/r03_dec_gemm_bench_h.log 2>&1; tail -20 gpurun_out/r03_dec_gemm_bench_h.log
timeout 900 python -m pytest tests/test_gpu_train.py tests/test_gpu_fullsize.py -m gpu -q -rf -k "not cfg4 and not cfg5 and not swin" > gpurun_out/r03_pytest_h.log 2>&1; tail -8 gpurun_out/r03_pytest_h.log
B="--no-cpu-baseline --no-roofline --no-parity --no-parity-mode --no-ref-batch --no-torch-baseline"
timeout 300 python bench.py --steps 6 --warmup 2 $B > gpurun_out/r03_bench_h_bf16.log 2>&1; tail -c 900 gpurun_out/r03_bench_h_bf16.log | head -c 500; echo
