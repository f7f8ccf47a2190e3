#!/bin/bash
# GPU session of the moment (overwritten per session; history in git).  Run as: gpurun --timeout N -- bash tools/gpu_session.sh
# round 6, FINAL-8 (last tree): the whole -m gpu suite + smoke()
cd "$GRAFT_REPO_ROOT" || exit 1
REPO="$GRAFT_REPO_ROOT"; O=$REPO/gpurun_out; mkdir -p $O
rm -f $O/parity_report.jsonl
timeout 2400 python -m pytest tests/ -q -m gpu > $O/r06_pytest_as_full.log 2>&1; echo "full suite rc $?"; tail -2 $O/r06_pytest_as_full.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $O/r06_smoke_as.log 2>&1; echo "smoke rc $?"; tail -1 $O/r06_smoke_as.log | cut -c1-200
