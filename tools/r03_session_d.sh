#!/bin/bash
# r03 GPU session D: op-level parity of the new kernels (x3 flash attention on planes, token-grouped chan_logits_bwd, batched colsum,
# Swin backward kernels), the two previously failing tests, x3f / bf16 bench lines, bf16 forward + x3f step profiles, PMC traffic passes.
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
REPO="$GRAFT_REPO_ROOT"
rm -f gpurun_out/parity_report.jsonl
timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -q -rf > gpurun_out/r03_pytest_d_ops.log 2>&1; tail -6 gpurun_out/r03_pytest_d_ops.log
timeout 900 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_ddp.py tests/test_gpu_train.py -m gpu -q -rf -k "flash_attention or cfg4 or torchrun or x3f or reproducible or swin" > gpurun_out/r03_pytest_d_sel.log 2>&1; tail -8 gpurun_out/r03_pytest_d_sel.log
cp gpurun_out/parity_report.jsonl gpurun_out/r03_parity_report_d.jsonl 2>/dev/null
SECONDS=0
timeout 900 python bench.py > gpurun_out/r03_bench_d_driver_style.log 2>&1
echo "bench.py (default flags) took ${SECONDS}s"; tail -c 3000 gpurun_out/r03_bench_d_driver_style.log; echo
(cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_d1 -o fwd -- python "$REPO/tools/fwd_bench.py" --prec bf16 --batch 63 --iters 3 --warmup 1 > "$REPO/gpurun_out/r03_prof_d1.log" 2>&1; python "$REPO/tools/prof_summary.py" /tmp/prof_d1 4 > "$REPO/gpurun_out/r03_fwd_bf16_b63_d.txt" 2>&1)
head -30 gpurun_out/r03_fwd_bf16_b63_d.txt | cut -c1-150
(cd /tmp && export TMPDIR=/tmp && timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_d2 -o train -- python "$REPO/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-parity --no-parity-mode --no-ref-batch --no-torch-baseline > "$REPO/gpurun_out/r03_prof_d2.log" 2>&1; python "$REPO/tools/prof_summary.py" /tmp/prof_d2 5 > "$REPO/gpurun_out/r03_train_ns6_b63_d.txt" 2>&1)
head -36 gpurun_out/r03_train_ns6_b63_d.txt | cut -c1-150
B="--steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-parity --no-parity-mode --no-ref-batch --no-torch-baseline"
(cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/pmc_f -o f -- python "$REPO/bench.py" $B > /dev/null 2>&1
 timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/pmc_w -o w -- python "$REPO/bench.py" $B > /dev/null 2>&1
 python "$REPO/tools/pmc_traffic.py" /tmp/pmc_f /tmp/pmc_w 'gemm_dma_kernel<1>' > "$REPO/gpurun_out/r03_pmc_traffic_d.json" 2>&1)
cat gpurun_out/r03_pmc_traffic_d.json
