#!/bin/bash
# GPU session of the moment (overwritten per session; history in git).  Run as: gpurun --timeout N -- bash tools/gpu_session.sh
cd "$GRAFT_REPO_ROOT" || exit 1
REPO="$GRAFT_REPO_ROOT"; O=$REPO/gpurun_out; mkdir -p $O
timeout 500 python tools/trajectory_fullsize.py 200 8 2e-4 > $O/r05_trajectory_fullsize.log 2> $O/r05_trajectory_fullsize.err; echo rc $?
tail -c 1800 $O/r05_trajectory_fullsize.log; tail -3 $O/r05_trajectory_fullsize.err
