#!/bin/bash
# r03 GPU session F: LDS-DMA issue schedule experiment (MTT_DMA_SCHED=1: DMA issues interleaved with the C-phase MFMAs) on a library build,
# bn_rowwise 4-row unroll check + step time.
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
{
  python tools/gemm_bench.py --lib tools/_exp/libmtt_hip_s1.so --split
  python tools/gemm_bench.py --split
  python tools/gemm_bench.py --lib tools/_exp/libmtt_hip_s1.so
} > gpurun_out/r03_gemm_bench_f_dma_sched.log 2>&1
grep -v "^$\|amdgpu.ids" gpurun_out/r03_gemm_bench_f_dma_sched.log | cut -c1-230
timeout 300 python -m pytest tests/test_gpu_ops.py -m gpu -q -k "bn_" > gpurun_out/r03_pytest_f.log 2>&1; tail -3 gpurun_out/r03_pytest_f.log
timeout 400 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-roofline --no-parity --no-parity-mode --no-ref-batch --no-torch-baseline > gpurun_out/r03_bench_f.log 2>&1
tail -c 700 gpurun_out/r03_bench_f.log | head -c 330; echo
