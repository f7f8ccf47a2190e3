#!/bin/bash
# GPU session of the moment (overwritten per session; history in git).  Run as: gpurun --timeout N -- bash tools/gpu_session.sh
cd "$GRAFT_REPO_ROOT" || exit 1
REPO="$GRAFT_REPO_ROOT"; O=$REPO/gpurun_out; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -k "conv3" > $O/r05_pytest_i_conv.log 2>&1; echo "pytest rc $?" >> $O/r05_pytest_i_conv.log
tail -4 $O/r05_pytest_i_conv.log
timeout 300 python tools/conv_bench.py 16 > $O/r05_conv_bench_i_ringc_edge.log 2>&1; cat $O/r05_conv_bench_i_ringc_edge.log | grep conv3x3
timeout 900 python -m pytest tests/test_gpu_train.py tests/test_gpu_model.py -x -q -k "not swin and not trajectory" > $O/r05_pytest_i_models.log 2>&1; echo "pytest rc $?" >> $O/r05_pytest_i_models.log
tail -4 $O/r05_pytest_i_models.log
timeout 400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-torch-baseline --no-ref-batch --no-x3-mode --no-parity > $O/r05_bench_i_ringc_edge.log 2> $O/r05_bench_i_ringc_edge.err; echo "bench rc $?"
timeout 420 python bench.py --config cfg4 --steps 8 --warmup 2 --no-cpu-baseline --no-torch-baseline --no-ref-batch --no-x3-mode --no-parity > $O/r05_bench_i_cfg4.log 2> $O/r05_bench_i_cfg4.err; echo "cfg4 rc $?"
python - <<'PY'
import json
for f in ('r05_bench_i_ringc_edge', 'r05_bench_i_cfg4'):
    l = [x for x in open(f'gpurun_out/{f}.log') if x.startswith('{')][-1]
    d = json.loads(l)
    fm = d.get('fast_mode') or {}
    print(f, 'x3f', d['value'], 'img/s', d['ms_per_step'], 'ms fwd', d['fwd_ms_per_img'], '| bf16', fm.get('images_per_s'), fm.get('ms_per_step'), fm.get('fwd_ms_per_img'))
PY
