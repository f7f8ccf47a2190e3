#!/bin/bash
# GPU session B of round 2: op-level parity of the new kernels, A/B micro-benchmarks, bench line + kernel profile, remaining parity tests.
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
rm -f gpurun_out/parity_report.jsonl
timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_losses_golden.py -m gpu -q -rf > gpurun_out/r02_pytest_b_ops.log 2>&1
tail -15 gpurun_out/r02_pytest_b_ops.log
timeout 300 python tools/gemm_bench.py 5 > gpurun_out/r02_gemm_bench_b.log 2>&1; cat gpurun_out/r02_gemm_bench_b.log
timeout 300 python tools/conv_bench.py 16 > gpurun_out/r02_conv_bench_b.log 2>&1; cat gpurun_out/r02_conv_bench_b.log
timeout 600 python bench.py --steps 6 --warmup 2 > gpurun_out/r02_bench_b.log 2>&1
tail -1 gpurun_out/r02_bench_b.log
(cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_b -o train -- python "$GRAFT_REPO_ROOT/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-parity --no-ref-batch > "$GRAFT_REPO_ROOT/gpurun_out/r02_prof_b.log" 2>&1; python "$GRAFT_REPO_ROOT/tools/prof_summary.py" /tmp/prof_b 4 > "$GRAFT_REPO_ROOT/gpurun_out/r02_prof_b.txt" 2>&1)
head -24 gpurun_out/r02_prof_b.txt
timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_train.py tests/test_gpu_fullsize.py "tests/test_gpu_configs.py::test_baseline_config_forward_matches_oracle[cfg5-x3]" "tests/test_gpu_configs.py::test_baseline_config_forward_matches_oracle[cfg5-bf16]" -m gpu -q -rf --durations=10 > gpurun_out/r02_pytest_b_model.log 2>&1
tail -30 gpurun_out/r02_pytest_b_model.log
