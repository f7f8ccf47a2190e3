#!/bin/bash
# r03 GPU session U: x3f training-step profile on the final code
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
REPO="$GRAFT_REPO_ROOT"
B="--no-cpu-baseline --no-roofline --no-parity --no-parity-mode --no-ref-batch --no-torch-baseline"
(cd /tmp && export TMPDIR=/tmp && timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_u -o train -- python "$REPO/bench.py" --prec x3f --steps 2 --warmup 1 --no-fwd $B > "$REPO/gpurun_out/r03_prof_u.log" 2>&1; python "$REPO/tools/prof_summary.py" /tmp/prof_u 4 > "$REPO/gpurun_out/r03_train_ns6_b63_x3f_u.txt" 2>&1)
head -48 gpurun_out/r03_train_ns6_b63_x3f_u.txt | cut -c1-165
grep -A30 "GEMM-family" gpurun_out/r03_train_ns6_b63_x3f_u.txt | cut -c1-120 | head -34
