#!/bin/bash
# GPU session Y: FULL -m gpu suite on the final kernels (LDS-DMA attention fwd/bwd, graph capture), smoke, driver-style bench line, kernel
# profile of the step, bench lines of cfg4 (InvPT) and cfg5 (N = 8194).
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
rm -f gpurun_out/parity_report.jsonl
timeout 900 python -m pytest tests -m gpu -q -rf --durations=6 > gpurun_out/r02_pytest_y_full.log 2>&1
tail -14 gpurun_out/r02_pytest_y_full.log
cp gpurun_out/parity_report.jsonl gpurun_out/r02_parity_report_y_full_suite.jsonl 2>/dev/null
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r02_smoke_y.log 2>&1; tail -2 gpurun_out/r02_smoke_y.log
timeout 500 python bench.py > gpurun_out/r02_bench_y_driver_style.log 2>&1
tail -c 600 gpurun_out/r02_bench_y_driver_style.log; echo
REPO="$GRAFT_REPO_ROOT"
(cd /tmp && export TMPDIR=/tmp && timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_y -o train -- python "$REPO/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-parity --no-ref-batch > "$REPO/gpurun_out/r02_prof_y.log" 2>&1; python "$REPO/tools/prof_summary.py" /tmp/prof_y 4 > "$REPO/gpurun_out/r02_train_ns6_b63_y.txt" 2>&1)
head -24 gpurun_out/r02_train_ns6_b63_y.txt | cut -c1-150
for c in cfg4 cfg5; do
  timeout 240 python bench.py --config $c --steps 4 --warmup 1 --no-cpu-baseline --no-parity --no-roofline --no-ref-batch > gpurun_out/r02_bench_y_$c.log 2>&1
  tail -c 400 gpurun_out/r02_bench_y_$c.log | head -c 300; echo
done
