#!/bin/bash
# GPU session K: FULL -m gpu suite on the final kernels, driver-style bench line, batch sweep, PMC traffic of the dominant kernel.
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
rm -f gpurun_out/parity_report.jsonl
timeout 1100 python -m pytest tests -m gpu -q -rf --durations=8 > gpurun_out/r02_pytest_k_full.log 2>&1
tail -14 gpurun_out/r02_pytest_k_full.log
timeout 400 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r02_smoke_k.log 2>&1; tail -2 gpurun_out/r02_smoke_k.log
timeout 500 python bench.py --steps 20 --warmup 5 > gpurun_out/r02_bench_k.log 2>&1
tail -c 600 gpurun_out/r02_bench_k.log; echo
for b in 95 127; do timeout 300 python bench.py --batch $b --steps 4 --warmup 2 --no-cpu-baseline --no-parity --no-ref-batch --no-roofline 2>&1 | tail -1 | cut -c1-260 | tee gpurun_out/r02_bench_k_b$b.log; done
REPO="$GRAFT_REPO_ROOT"
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-parity --no-ref-batch"
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/pmc_f -o f -- python "$REPO/bench.py" $ARGS > "$REPO/gpurun_out/r02_pmc_f.log" 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/pmc_w -o w -- python "$REPO/bench.py" $ARGS > "$REPO/gpurun_out/r02_pmc_w.log" 2>&1
python "$REPO/tools/pmc_traffic.py" /tmp/pmc_f /tmp/pmc_w "gemm_dma_kernel<256, false, 0>" > "$REPO/gpurun_out/pmc_traffic.json" 2>&1
cut -c1-240 "$REPO/gpurun_out/pmc_traffic.json"
