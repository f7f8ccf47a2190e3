#!/bin/bash
# GPU session of the moment (overwritten per session; history in git).  Run as: gpurun --timeout N -- bash tools/gpu_session.sh
# round 6, session 4: the hand-interleaved pipelined x3 flash forward: parity cases + A/B against the three-phase kernel.
cd "$GRAFT_REPO_ROOT" || exit 1
REPO="$GRAFT_REPO_ROOT"; O=$REPO/gpurun_out; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_fullsize.py -x -q -m gpu -k "attn or attention" > $O/r06_pytest_d_attn.log 2>&1; echo "pytest rc $?"; tail -3 $O/r06_pytest_d_attn.log
timeout 300 python tools/attn_x3_bench.py > $O/r06_attn_x3_bench_d.log 2>&1; tail -2 $O/r06_attn_x3_bench_d.log
timeout 300 python tools/attn_x3_bench.py 4 8194 16 2 >> $O/r06_attn_x3_bench_d.log 2>&1; tail -2 $O/r06_attn_x3_bench_d.log
timeout 300 python tools/attn_x3_bench.py 8 1024 16 6 >> $O/r06_attn_x3_bench_d.log 2>&1; tail -2 $O/r06_attn_x3_bench_d.log
