#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
REPO="$GRAFT_REPO_ROOT"; O=$REPO/gpurun_out; mkdir -p $O
B="--no-cpu-baseline --no-roofline --no-parity --no-fast-mode --no-ref-batch --no-torch-baseline"
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_j -o train -- python $REPO/bench.py --prec x3f --steps 2 --warmup 1 --no-fwd $B > $O/r04_prof_j.log 2>&1
python $REPO/tools/prof_summary.py /tmp/prof_j 4 > $O/r04_train_ns6_b63_x3f_j.txt 2>&1
head -64 $O/r04_train_ns6_b63_x3f_j.txt | cut -c1-150
grep -A25 "GEMM-family" $O/r04_train_ns6_b63_x3f_j.txt | cut -c1-110
