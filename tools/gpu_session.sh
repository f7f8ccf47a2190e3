#!/bin/bash
# GPU session of the moment (overwritten per session; history in git).  Run as: gpurun --timeout N -- bash tools/gpu_session.sh
cd "$GRAFT_REPO_ROOT" || exit 1
REPO="$GRAFT_REPO_ROOT"; O=$REPO/gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_train.py tests/test_gpu_fullsize.py tests/test_gpu_ddp.py -x -q -k "not swin and not invpt and not cfg4 and not trajectory" > $O/r05_pytest_m_train.log 2>&1; tail -3 $O/r05_pytest_m_train.log
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
    _orig = ap.FuseTailFn.apply
    def _three_nodes(y0, geo1, tags, F, prec, training, bns, *params):
        Z = len(params) // 6
        cw, cb, bg, bb, w4, b4 = (params[i * Z:(i + 1) * Z] for i in range(6))
        y1 = ap.Conv3x3Fn.apply(y0, geo1, prec, tags[0], *cw, *cb)
        y1 = ap._bn_act(y1, bns, F, ap.ACT_GELU, training)
        return ap.BLinearFn.apply(y1, F, 'plain', None, None, prec, tags[1], None, *w4, *b4)
    ap.FuseTailFn.apply = staticmethod(_three_nodes)
bench.main()
PY
done
