#!/bin/bash
# GPU session L: taps-first ConvHead (upconv kernels) + deterministic colsum: op parity, model / gradient parity, micro-benchmark, bench A/B,
# full-size parity of the configs that use ConvHeads, kernel profile.
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
rm -f gpurun_out/parity_report.jsonl
timeout 300 python -m pytest tests/test_gpu_ops.py -m gpu -q -rf -k "upconv or colsum" > gpurun_out/r02_pytest_l_ops.log 2>&1
tail -6 gpurun_out/r02_pytest_l_ops.log
timeout 400 python -m pytest tests/test_gpu_model.py tests/test_gpu_train.py -m gpu -q -rf > gpurun_out/r02_pytest_l_model.log 2>&1
tail -4 gpurun_out/r02_pytest_l_model.log
timeout 300 python tools/upconv_bench.py 63 > gpurun_out/r02_upconv_bench_l.log 2>&1
cat gpurun_out/r02_upconv_bench_l.log | tail -12
timeout 500 python bench.py --steps 10 --warmup 3 > gpurun_out/r02_bench_l.log 2>&1
tail -c 1500 gpurun_out/r02_bench_l.log | cut -c1-1500; echo
timeout 300 python bench.py --steps 6 --warmup 2 --no-fuse-upsample --no-cpu-baseline --no-parity --no-ref-batch --no-roofline 2>&1 | tail -1 | cut -c1-330 | tee gpurun_out/r02_bench_l_unfused.log
timeout 700 python -m pytest tests/test_gpu_configs.py tests/test_gpu_fullsize.py -m gpu -q -rf -k "ns6 or cfg2 or cfg3 or training_step" > gpurun_out/r02_pytest_l_fullsize.log 2>&1
tail -6 gpurun_out/r02_pytest_l_fullsize.log
cp gpurun_out/parity_report.jsonl gpurun_out/r02_parity_report_l.jsonl 2>/dev/null
(cd /tmp && export TMPDIR=/tmp && timeout 500 rocprofv3 --kernel-trace --stats -d /tmp/prof_l -o train -- python "$GRAFT_REPO_ROOT/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-parity --no-ref-batch > "$GRAFT_REPO_ROOT/gpurun_out/r02_prof_l.log" 2>&1; python "$GRAFT_REPO_ROOT/tools/prof_summary.py" /tmp/prof_l 4 > "$GRAFT_REPO_ROOT/gpurun_out/r02_prof_l.txt" 2>&1)
head -45 gpurun_out/r02_prof_l.txt
