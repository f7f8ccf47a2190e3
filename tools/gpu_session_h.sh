#!/bin/bash
# GPU session H: specialised interior-tile epilogue: parity of every GEMM case, ablation A/B, micro-bench, bench line, model tests.
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_gpu_ops.py -m gpu -q -rf -k "gemm" > gpurun_out/r02_pytest_h_ops.log 2>&1
tail -6 gpurun_out/r02_pytest_h_ops.log
timeout 300 python tools/gemm_ablate.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r02_gemm_ablate_h.log; cat gpurun_out/r02_gemm_ablate_h.log
timeout 300 python tools/conv_bench.py 16 2>&1 | grep -v amdgpu.ids > gpurun_out/r02_conv_bench_h.log; cat gpurun_out/r02_conv_bench_h.log
timeout 400 python bench.py --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/r02_bench_h.log 2>&1
tail -c 1300 gpurun_out/r02_bench_h.log
timeout 600 python -m pytest tests/test_gpu_model.py tests/test_gpu_train.py tests/test_gpu_fullsize.py -m gpu -q -rf > gpurun_out/r02_pytest_h_model.log 2>&1
tail -5 gpurun_out/r02_pytest_h_model.log
