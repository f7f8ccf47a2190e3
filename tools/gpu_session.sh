#!/bin/bash
# GPU session of the moment (overwritten per session; history in git).  Run as: gpurun --timeout N -- bash tools/gpu_session.sh
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out; mkdir -p $O
export TMPDIR=/tmp
python tools/gemm_bench.py --split --rounds 3 > $O/r04_gemm_a_ring_s4.log 2>&1
for v in old s5 s4n2 s5n2 s4n0; do
  python tools/gemm_bench.py --split --rounds 3 --no-check --lib build/variants/libmtt_$v.so > $O/r04_gemm_a_$v.log 2>&1
done
tail -n 30 $O/r04_gemm_a_*.log
timeout 900 python bench.py --steps 5 --warmup 2 --no-torch-baseline --no-ref-batch > $O/r04_bench_a.log 2>&1
tail -c 6000 $O/r04_bench_a.log
