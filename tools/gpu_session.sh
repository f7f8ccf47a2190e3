#!/bin/bash
# GPU session of the moment (overwritten per session; history in git).  Run as: gpurun --timeout N -- bash tools/gpu_session.sh
cd "$GRAFT_REPO_ROOT" || exit 1
REPO="$GRAFT_REPO_ROOT"; O=$REPO/gpurun_out; mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_ops.py -x -q -k "dma128" > $O/r05_pytest_j_ops.log 2>&1; tail -3 $O/r05_pytest_j_ops.log
timeout 600 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_train.py -x -q -k "ns6 or fullsize_backward or gradients" > $O/r05_pytest_j_train.log 2>&1; tail -3 $O/r05_pytest_j_train.log
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $O/r05_smoke_j.log 2>&1; tail -2 $O/r05_smoke_j.log
Q="--steps 10 --warmup 3 --no-cpu-baseline --no-torch-baseline --no-ref-batch --no-x3-mode --no-fast-mode --no-parity --no-fwd --no-roofline"
for mode in new old new old; do
python - $mode $Q <<'PY' 2>/dev/null | python -c "import sys, json; [print(sys.argv[1], json.loads(l)['value'], json.loads(l)['ms_per_step']) for l in sys.stdin if l.startswith('{')]" $mode
import sys, importlib
sys.path.insert(0, '.')
mode = sys.argv[1]
sys.argv = ['bench.py'] + sys.argv[2:]
import bench
ap = importlib.import_module("multi-task-transformer_amd.autograd_path")
if mode == 'old':
    ap.HEAD_DGRAD_DMA = False
    ap.WGRAD_MAX_SLICES, ap.WGRAD_TARGET_WGS = 64, 256
bench.main()
PY
done
