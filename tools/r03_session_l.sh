#!/bin/bash
# r03 GPU session L: training-step and forward profiles after the decoder / launch work
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
REPO="$GRAFT_REPO_ROOT"
B="--no-cpu-baseline --no-roofline --no-parity --no-parity-mode --no-ref-batch --no-torch-baseline"
(cd /tmp && export TMPDIR=/tmp && timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_l -o train -- python "$REPO/bench.py" --steps 3 --warmup 1 --no-fwd $B > "$REPO/gpurun_out/r03_prof_l.log" 2>&1; python "$REPO/tools/prof_summary.py" /tmp/prof_l 5 > "$REPO/gpurun_out/r03_train_ns6_b63_l.txt" 2>&1)
head -64 gpurun_out/r03_train_ns6_b63_l.txt | cut -c1-170
(cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_l1 -o fwd -- python "$REPO/tools/fwd_bench.py" --prec bf16 --batch 63 --iters 3 --warmup 1 > "$REPO/gpurun_out/r03_prof_l1.log" 2>&1; python "$REPO/tools/prof_summary.py" /tmp/prof_l1 4 > "$REPO/gpurun_out/r03_fwd_bf16_b63_l.txt" 2>&1)
head -40 gpurun_out/r03_fwd_bf16_b63_l.txt | cut -c1-170
