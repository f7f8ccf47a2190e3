#!/bin/bash
# GPU session I: token-major weight-gradient kernel: parity cases, micro-benchmark vs the round-1 paths, model gradient tests, bench.
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_gpu_ops.py -m gpu -q -rf -k "gemm" > gpurun_out/r02_pytest_i_ops.log 2>&1
tail -15 gpurun_out/r02_pytest_i_ops.log
timeout 300 python tools/wgrad_bench.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r02_wgrad_bench_i.log; cat gpurun_out/r02_wgrad_bench_i.log
timeout 400 python -m pytest tests/test_gpu_train.py tests/test_gpu_fullsize.py -m gpu -q -rf > gpurun_out/r02_pytest_i_model.log 2>&1
tail -8 gpurun_out/r02_pytest_i_model.log
timeout 400 python bench.py --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/r02_bench_i.log 2>&1
tail -c 900 gpurun_out/r02_bench_i.log
