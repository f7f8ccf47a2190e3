#!/bin/bash
# GPU session of the moment (overwritten per session; history in git).  Run as: gpurun --timeout N -- bash tools/gpu_session.sh
cd "$GRAFT_REPO_ROOT" || exit 1
REPO="$GRAFT_REPO_ROOT"; O=$REPO/gpurun_out; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -k "attn" > $O/r05_pytest_d_attn.log 2>&1; echo "pytest rc $?" >> $O/r05_pytest_d_attn.log
tail -4 $O/r05_pytest_d_attn.log
for n in 1030 1024 150 8194; do b=63; [ $n = 8194 ] && b=4; timeout 200 python tools/attn_x3_bench.py $b $n 16 6 >> $O/r05_attn_x3_bench_d_ragged.log 2>&1; timeout 200 python tools/attn_x3_bench.py $b $n 16 6 --lib build/variants/libmtt_noragged.so >> $O/r05_attn_x3_bench_d_noragged.log 2>&1; done
grep -h "attention forward\|library" $O/r05_attn_x3_bench_d_ragged.log $O/r05_attn_x3_bench_d_noragged.log
timeout 600 python -m pytest tests/test_gpu_fullsize.py -x -q -k "attention or attn" > $O/r05_pytest_d_full.log 2>&1; tail -3 $O/r05_pytest_d_full.log
