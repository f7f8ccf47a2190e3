#!/bin/bash
# r03 GPU session G: one-launch weight re-packing (mtt_segcopy), bias gradients from the producers (GEMM epilogue column sums, rowscale cast
# + column sums), foreach BN running stats: op parity, training / model / graph tests, bf16 + x3f bench lines, launches per training step.
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
REPO="$GRAFT_REPO_ROOT"
timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -q -rf -x -k "colsum or segcopy or rowscale_cast or gemm_epi or gemm_dma" > gpurun_out/r03_pytest_g_ops.log 2>&1; tail -6 gpurun_out/r03_pytest_g_ops.log
timeout 1200 python -m pytest tests/test_gpu_train.py tests/test_gpu_model.py -m gpu -q -rf > gpurun_out/r03_pytest_g_train.log 2>&1; tail -8 gpurun_out/r03_pytest_g_train.log
B="--no-cpu-baseline --no-roofline --no-parity --no-parity-mode --no-ref-batch --no-torch-baseline"
timeout 300 python bench.py --steps 6 --warmup 2 $B > gpurun_out/r03_bench_g_bf16.log 2>&1; tail -c 1500 gpurun_out/r03_bench_g_bf16.log; echo
timeout 300 python bench.py --prec x3f --steps 4 --warmup 2 $B > gpurun_out/r03_bench_g_x3f.log 2>&1; tail -c 1200 gpurun_out/r03_bench_g_x3f.log; echo
(cd /tmp && export TMPDIR=/tmp && timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_g -o train -- python "$REPO/bench.py" --steps 3 --warmup 1 --no-fwd $B > "$REPO/gpurun_out/r03_prof_g.log" 2>&1; python "$REPO/tools/prof_summary.py" /tmp/prof_g 5 > "$REPO/gpurun_out/r03_train_ns6_b63_g.txt" 2>&1)
head -50 gpurun_out/r03_train_ns6_b63_g.txt | cut -c1-160
