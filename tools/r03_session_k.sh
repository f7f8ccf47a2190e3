#!/bin/bash
# r03 GPU session K: the encoder shapes with their real epilogues on the 128 x 128 two-workgroup kernel vs the 256 x 256 kernel
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 200 python tools/gemm_bench.py --variant 3 > gpurun_out/r03_gemm_bench_k_v3.log 2>&1; cat gpurun_out/r03_gemm_bench_k_v3.log
timeout 200 python tools/gemm_bench.py --variant 4 > gpurun_out/r03_gemm_bench_k_v4.log 2>&1; cat gpurun_out/r03_gemm_bench_k_v4.log
