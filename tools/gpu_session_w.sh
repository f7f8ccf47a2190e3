#!/bin/bash
# GPU session W: entry-point suite (incl. attention variants) + driver-style bench line with the LDS-DMA attention kernels and the new weight-gradient slicing.
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_ops.py -q -x -rf > gpurun_out/r02_pytest_w_ops.log 2>&1
tail -4 gpurun_out/r02_pytest_w_ops.log
timeout 500 python bench.py --no-cpu-baseline > gpurun_out/r02_bench_w.log 2>&1
tail -c 1500 gpurun_out/r02_bench_w.log; echo
