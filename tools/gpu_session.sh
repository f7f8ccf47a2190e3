#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out; mkdir -p $O
timeout 2400 python -m pytest tests/test_gpu_fullsize.py -m gpu -x -q -k swinb -s > $O/r04_pytest_k_swinb.log 2>&1; tail -25 $O/r04_pytest_k_swinb.log | cut -c1-400
grep cs_swinB $O/parity_report.jsonl | tail -3 | cut -c1-600
