#!/bin/bash
# GPU session of the moment (overwritten per session; history in git).  Run as: gpurun --timeout N -- bash tools/gpu_session.sh
# round 5: the whole -m gpu suite and smoke() on the FINAL tree.
cd "$GRAFT_REPO_ROOT" || exit 1
REPO="$GRAFT_REPO_ROOT"; O=$REPO/gpurun_out; mkdir -p $O
rm -f $O/parity_report.jsonl
timeout 900 python -m pytest tests -m gpu -x -q > $O/r05_pytest_final.log 2>&1; echo "pytest rc $?" >> $O/r05_pytest_final.log
tail -5 $O/r05_pytest_final.log
cp $O/parity_report.jsonl $O/r05_parity_report_final.jsonl 2>/dev/null
timeout 200 python -c "import __graft_entry__ as g; g.build(); g.smoke()" > $O/r05_smoke_final.log 2>&1; tail -1 $O/r05_smoke_final.log
