#!/bin/bash
# GPU session of the moment (overwritten per session; history in git).  Run as: gpurun --timeout N -- bash tools/gpu_session.sh
cd "$GRAFT_REPO_ROOT" || exit 1
REPO="$GRAFT_REPO_ROOT"; O=$REPO/gpurun_out; mkdir -p $O
rm -f $O/parity_report.jsonl
timeout 900 python -m pytest tests -m gpu -x -q > $O/r05_pytest_a.log 2>&1; echo "pytest rc $?" >> $O/r05_pytest_a.log
tail -5 $O/r05_pytest_a.log
cp $O/parity_report.jsonl $O/r05_parity_report_a.jsonl 2>/dev/null
timeout 600 python bench.py --steps 20 --warmup 5 > $O/r05_bench_a.log 2> $O/r05_bench_a.err; echo "bench rc $?"
tail -c 1500 $O/r05_bench_a.log
