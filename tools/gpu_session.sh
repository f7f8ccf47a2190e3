#!/bin/bash
# GPU session of the moment (overwritten per session; history in git).  Run as: gpurun --timeout N -- bash tools/gpu_session.sh
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out; mkdir -p $O
export TMPDIR=/tmp
python tools/gemm_bench.py --rounds 3 --lib build/variants/libmtt_pair.so > $O/r04_gemm_d_pair.log 2>&1
python tools/gemm_bench.py --rounds 3 --no-check --lib build/variants/libmtt_old.so > $O/r04_gemm_d_old.log 2>&1
python tools/gemm_trace.py --lib build/variants/libmtt_pairtrace.so > $O/r04_gemm_trace_d_pair.log 2>&1
cat $O/r04_gemm_d_pair.log $O/r04_gemm_d_old.log $O/r04_gemm_trace_d_pair.log
