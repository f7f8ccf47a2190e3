#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out; mkdir -p $O
python tools/headpred_bench.py > $O/r04_headpred_bench_t.log 2>&1; cat $O/r04_headpred_bench_t.log
