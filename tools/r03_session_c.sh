#!/bin/bash
# r03 GPU session C: FULL -m gpu suite (determinism, Swin backward kernels, cfg4 training step, NMS 4096, x3f ...), smoke, the driver-style
# bench line with every leg (parity vs oracle, parity_mode, torch baseline, cpu baseline).
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
rm -f gpurun_out/parity_report.jsonl
timeout 1500 python -m pytest tests -m gpu -q -rf --durations=8 > gpurun_out/r03_pytest_c_full.log 2>&1
tail -25 gpurun_out/r03_pytest_c_full.log
cp gpurun_out/parity_report.jsonl gpurun_out/r03_parity_report_c_full_suite.jsonl 2>/dev/null
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r03_smoke_c.log 2>&1; tail -2 gpurun_out/r03_smoke_c.log
SECONDS=0
timeout 900 python bench.py > gpurun_out/r03_bench_c_driver_style.log 2>&1
echo "bench.py (default flags) took ${SECONDS}s"
tail -c 4000 gpurun_out/r03_bench_c_driver_style.log; echo
