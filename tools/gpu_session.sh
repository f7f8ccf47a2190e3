#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out; mkdir -p $O
python tools/gemm_bench.py --split --rounds 5 --no-check > $O/r04_gemm_w_gm4.log 2>&1
for v in gm2 gm8 gm16; do python tools/gemm_bench.py --split --rounds 5 --no-check --lib build/variants/libmtt_$v.so > $O/r04_gemm_w_$v.log 2>&1; done
python tools/gemm_bench.py --split --rounds 5 --no-check > $O/r04_gemm_w_gm4b.log 2>&1
grep -h "library\|split x3" $O/r04_gemm_w_*.log | cut -c1-230
