#!/bin/bash
# GPU session U: FULL -m gpu suite on the final kernels, smoke, driver-style bench line, kernel profile of the step, PMC traffic of the
# dominant kernel (separate FETCH_SIZE / WRITE_SIZE passes).
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
rm -f gpurun_out/parity_report.jsonl
timeout 1300 python -m pytest tests -m gpu -q -rf --durations=8 > gpurun_out/r02_pytest_u_full.log 2>&1
tail -16 gpurun_out/r02_pytest_u_full.log
cp gpurun_out/parity_report.jsonl gpurun_out/r02_parity_report_u_full_suite.jsonl 2>/dev/null
timeout 400 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r02_smoke_u.log 2>&1; tail -2 gpurun_out/r02_smoke_u.log
REPO="$GRAFT_REPO_ROOT"
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-parity --no-ref-batch"
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/pmc_f -o f -- python "$REPO/bench.py" $ARGS > "$REPO/gpurun_out/r02_pmc_u_f.log" 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/pmc_w -o w -- python "$REPO/bench.py" $ARGS > "$REPO/gpurun_out/r02_pmc_u_w.log" 2>&1
python "$REPO/tools/pmc_traffic.py" /tmp/pmc_f /tmp/pmc_w "gemm_dma_kernel<256, false, 7>" > "$REPO/gpurun_out/pmc_traffic.json" 2>&1
cut -c1-260 "$REPO/gpurun_out/pmc_traffic.json"
cp "$REPO/gpurun_out/pmc_traffic.json" "$REPO/profiles/pmc_traffic.json"
cd "$REPO"
timeout 500 python bench.py > gpurun_out/r02_bench_u_driver_style.log 2>&1
tail -c 700 gpurun_out/r02_bench_u_driver_style.log; echo
(cd /tmp && timeout 500 rocprofv3 --kernel-trace --stats -d /tmp/prof_u -o train -- python "$REPO/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-parity --no-ref-batch > "$REPO/gpurun_out/r02_prof_u.log" 2>&1; python "$REPO/tools/prof_summary.py" /tmp/prof_u 4 > "$REPO/gpurun_out/r02_train_ns6_b63_u.txt" 2>&1)
head -30 gpurun_out/r02_train_ns6_b63_u.txt
