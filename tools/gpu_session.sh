#!/bin/bash
# GPU session of the moment (overwritten per session; history in git).  Run as: gpurun --timeout N -- bash tools/gpu_session.sh
cd "$GRAFT_REPO_ROOT" || exit 1
REPO="$GRAFT_REPO_ROOT"; O=$REPO/gpurun_out; mkdir -p $O
timeout 400 python tools/swin_bench.py x3f 2,4,8 train > $O/r05_swin_bench_k_x3f_train.log 2>&1; grep swinB $O/r05_swin_bench_k_x3f_train.log
timeout 300 python tools/swin_bench.py bf16 2,8 train > $O/r05_swin_bench_k_bf16_train.log 2>&1; grep swinB $O/r05_swin_bench_k_bf16_train.log
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_k -o k -- python $REPO/tools/swin_bench.py x3f 4 train > $O/r05_prof_k_run.log 2>&1
python $REPO/tools/prof_summary.py /tmp/prof_k 1 > $O/r05_swin_train_b4_x3f_k.txt 2>&1
head -40 $O/r05_swin_train_b4_x3f_k.txt
