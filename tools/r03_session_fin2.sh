#!/bin/bash
# r03 GPU: forward-only and training-step profiles of the final commit (bf16)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
REPO="$GRAFT_REPO_ROOT"
B="--no-cpu-baseline --no-roofline --no-parity --no-parity-mode --no-ref-batch --no-torch-baseline"
(cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_f1 -o fwd -- python "$REPO/tools/fwd_bench.py" --prec bf16 --batch 63 --iters 3 --warmup 1 > "$REPO/gpurun_out/r03_prof_f1.log" 2>&1; python "$REPO/tools/prof_summary.py" /tmp/prof_f1 4 > "$REPO/gpurun_out/r03_fwd_bf16_b63_final.txt" 2>&1)
head -24 gpurun_out/r03_fwd_bf16_b63_final.txt | cut -c1-150
(cd /tmp && export TMPDIR=/tmp && timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_f2 -o train -- python "$REPO/bench.py" --steps 3 --warmup 1 --no-fwd $B > "$REPO/gpurun_out/r03_prof_f2.log" 2>&1; python "$REPO/tools/prof_summary.py" /tmp/prof_f2 5 > "$REPO/gpurun_out/r03_train_ns6_b63_final.txt" 2>&1)
head -30 gpurun_out/r03_train_ns6_b63_final.txt | cut -c1-150
