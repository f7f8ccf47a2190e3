#!/bin/bash
# r03 GPU session N: cfg4 (InvPT ViT-L, 6 tasks) training-step profile
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
REPO="$GRAFT_REPO_ROOT"
B="--no-cpu-baseline --no-roofline --no-parity --no-parity-mode --no-ref-batch --no-torch-baseline"
(cd /tmp && export TMPDIR=/tmp && timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_n -o train -- python "$REPO/bench.py" --config cfg4 --steps 2 --warmup 1 --no-fwd $B > "$REPO/gpurun_out/r03_prof_n.log" 2>&1; python "$REPO/tools/prof_summary.py" /tmp/prof_n 4 > "$REPO/gpurun_out/r03_train_cfg4_b32_n.txt" 2>&1)
head -75 gpurun_out/r03_train_cfg4_b32_n.txt | cut -c1-175
