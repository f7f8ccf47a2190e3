#!/bin/bash
# r03 GPU session B: the split-plane (x3f) kernels — op-level parity, model-level parity (forward 1e-3, gradients), bench + profile of the
# x3f step, and the bf16 bench line after the gemm.hip clean-up (no regression check).
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
REPO="$GRAFT_REPO_ROOT"
rm -f gpurun_out/parity_report.jsonl
timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -q -rf > gpurun_out/r03_pytest_b_ops.log 2>&1; tail -15 gpurun_out/r03_pytest_b_ops.log
timeout 600 python -m pytest tests/test_gpu_train.py -m gpu -q -rf -k "x3f or bf16_are" > gpurun_out/r03_pytest_b_train.log 2>&1; tail -8 gpurun_out/r03_pytest_b_train.log
timeout 900 python -m pytest tests/test_gpu_configs.py -m gpu -q -rf -k "ns6 or cfg1 or cfg2" > gpurun_out/r03_pytest_b_configs.log 2>&1; tail -8 gpurun_out/r03_pytest_b_configs.log
timeout 900 python -m pytest tests/test_gpu_fullsize.py -m gpu -q -rf -k "training_step or dma_gemm" > gpurun_out/r03_pytest_b_fullsize.log 2>&1; tail -8 gpurun_out/r03_pytest_b_fullsize.log
cp gpurun_out/parity_report.jsonl gpurun_out/r03_parity_report_b.jsonl 2>/dev/null
timeout 400 python bench.py --prec x3f --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-parity --no-ref-batch --no-torch-baseline > gpurun_out/r03_bench_b_x3f.log 2>&1
tail -c 900 gpurun_out/r03_bench_b_x3f.log; echo
timeout 400 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-parity --no-ref-batch --no-torch-baseline > gpurun_out/r03_bench_b_bf16.log 2>&1
tail -c 1500 gpurun_out/r03_bench_b_bf16.log | head -c 900; echo
(cd /tmp && export TMPDIR=/tmp && timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_b -o train -- python "$REPO/bench.py" --prec x3f --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-parity --no-ref-batch --no-torch-baseline > "$REPO/gpurun_out/r03_prof_b.log" 2>&1; python "$REPO/tools/prof_summary.py" /tmp/prof_b 3 > "$REPO/gpurun_out/r03_train_ns6_b63_x3f_b.txt" 2>&1)
head -34 gpurun_out/r03_train_ns6_b63_x3f_b.txt | cut -c1-150
