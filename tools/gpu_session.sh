#!/bin/bash
# round 6, session 22: Swin-B x3f step profile with the matrix-core window attention
cd "$GRAFT_REPO_ROOT" || exit 1
REPO="$GRAFT_REPO_ROOT"; O=$REPO/gpurun_out; mkdir -p $O
B="--no-torch-baseline --no-cpu-baseline --no-ref-batch --no-x3-mode --no-fast-mode --no-parity --no-roofline"
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof_sw -o swin -- python $REPO/bench.py --config swinb --steps 2 --warmup 1 $B > $O/r06_prof_w_swin_run.log 2>&1; echo "prof swin rc $?"
cd $REPO
python tools/prof_summary.py /tmp/prof_sw 3 > $O/r06_train_swinb_b8_x3f_w.txt 2>&1
head -24 $O/r06_train_swinb_b8_x3f_w.txt | cut -c1-200
