#!/bin/bash
# r03 GPU session A: precision attribution table, DDP-on-device tests, x3 forward kernel profile, bench with the stock-torch baseline leg.
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
REPO="$GRAFT_REPO_ROOT"
timeout 600 python tools/prec_attribution.py --out gpurun_out/r03_prec_attribution.json > gpurun_out/r03_prec_attribution.txt 2>&1
tail -16 gpurun_out/r03_prec_attribution.txt
timeout 900 python -m pytest tests/test_gpu_ddp.py -m gpu -q -rf -x > gpurun_out/r03_pytest_a_ddp.log 2>&1
tail -30 gpurun_out/r03_pytest_a_ddp.log
(cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_a -o fwd -- python "$REPO/tools/fwd_bench.py" --prec x3 --batch 24 --iters 3 --warmup 1 > "$REPO/gpurun_out/r03_prof_a.log" 2>&1; python "$REPO/tools/prof_summary.py" /tmp/prof_a 4 > "$REPO/gpurun_out/r03_fwd_x3_b24_a.txt" 2>&1)
tail -2 gpurun_out/r03_prof_a.log
head -30 gpurun_out/r03_fwd_x3_b24_a.txt | cut -c1-160
timeout 700 python bench.py --steps 4 --warmup 1 --no-ref-batch --no-cpu-baseline > gpurun_out/r03_bench_a.log 2>&1
tail -c 2500 gpurun_out/r03_bench_a.log; echo
