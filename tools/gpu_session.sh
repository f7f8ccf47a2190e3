#!/bin/bash
# GPU session of the moment (overwritten per session; history in git).  Run as: gpurun --timeout N -- bash tools/gpu_session.sh
cd "$GRAFT_REPO_ROOT" || exit 1
REPO="$GRAFT_REPO_ROOT"; O=$REPO/gpurun_out; mkdir -p $O
timeout 1800 python -m pytest tests -m gpu -x -q > $O/r04_pytest_v.log 2>&1; tail -6 $O/r04_pytest_v.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/r04_smoke_v.log 2>&1; tail -2 $O/r04_smoke_v.log
