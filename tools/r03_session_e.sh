#!/bin/bash
# r03 GPU session E: tile-order (GROUP_M) sweep of the dominant GEMM on library builds; x3 GEMM on planes; x3f bench with the new flash kernel.
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
{
  python tools/gemm_bench.py --split
  for g in 2 6 8 16; do python tools/gemm_bench.py --lib tools/_exp/libmtt_hip_gm$g.so; done
  python tools/gemm_bench.py
} > gpurun_out/r03_gemm_bench_e_group_m.log 2>&1
cat gpurun_out/r03_gemm_bench_e_group_m.log | grep -v "^$" | cut -c1-200
timeout 400 python bench.py --prec x3f --steps 4 --warmup 1 --no-cpu-baseline --no-roofline --no-parity --no-parity-mode --no-ref-batch --no-torch-baseline > gpurun_out/r03_bench_e_x3f.log 2>&1
tail -c 700 gpurun_out/r03_bench_e_x3f.log | head -c 400; echo
