#!/bin/bash
# GPU session E: epilogue with prefetched residual / aux rows: parity of every GEMM epilogue case, tile-time ablation, bench.
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_gpu_ops.py -m gpu -q -rf -k "gemm" > gpurun_out/r02_pytest_e_ops.log 2>&1
tail -4 gpurun_out/r02_pytest_e_ops.log
timeout 300 python tools/gemm_ablate.py > gpurun_out/r02_gemm_ablate_e.log 2>&1; cat gpurun_out/r02_gemm_ablate_e.log
timeout 400 python bench.py --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/r02_bench_e.log 2>&1
tail -c 1200 gpurun_out/r02_bench_e.log
timeout 400 python -m pytest tests/test_gpu_model.py tests/test_gpu_train.py -m gpu -q -rf > gpurun_out/r02_pytest_e_model.log 2>&1
tail -4 gpurun_out/r02_pytest_e_model.log
