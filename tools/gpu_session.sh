#!/bin/bash
# round 6, session 38: per-step times of the first iterations (is bench.py's default warm-up of 2 steps enough?)
cd "$GRAFT_REPO_ROOT" || exit 1
REPO="$GRAFT_REPO_ROOT"; O=$REPO/gpurun_out; mkdir -p $O
timeout 900 python tools/step_times.py ns6 x3f 14 2>&1 | grep "^step" | tee $O/r06_step_times_ao_ns6.log
