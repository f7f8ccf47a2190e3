#!/bin/bash
# GPU session J: kernel profile of the bench command after the token-major wgrad kernel; op tests (TN cases).
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_gpu_ops.py -m gpu -q -rf -k "gemm_tn" > gpurun_out/r02_pytest_j_ops.log 2>&1
tail -3 gpurun_out/r02_pytest_j_ops.log
(cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_j -o train -- python "$GRAFT_REPO_ROOT/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-parity --no-ref-batch > "$GRAFT_REPO_ROOT/gpurun_out/r02_prof_j.log" 2>&1; python "$GRAFT_REPO_ROOT/tools/prof_summary.py" /tmp/prof_j 4 > "$GRAFT_REPO_ROOT/gpurun_out/r02_prof_j.txt" 2>&1)
head -50 gpurun_out/r02_prof_j.txt
