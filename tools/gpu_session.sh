#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
REPO="$GRAFT_REPO_ROOT"; O=$REPO/gpurun_out; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_configs.py tests/test_gpu_model.py tests/test_gpu_train.py -m gpu -x -q > $O/r04_pytest_p.log 2>&1; tail -4 $O/r04_pytest_p.log
timeout 900 python -m pytest tests/test_gpu_fullsize.py -m gpu -x -q -k "cfg4" > $O/r04_pytest_p2.log 2>&1; tail -3 $O/r04_pytest_p2.log
S="--steps 5 --warmup 2 --no-torch-baseline --no-ref-batch --no-x3-mode --no-cpu-baseline --no-fast-mode"
for c in cfg4 cfg2 cfg3; do timeout 900 python bench.py --config $c $S > $O/r04_bench_p_$c.log 2>&1; python - <<P
import json
l=[x for x in open('gpurun_out/r04_bench_p_$c.log') if x.startswith('{')]
if l:
    d=json.loads(l[-1]); print('$c x3f', d['value'], d['ms_per_step'], d['fwd_ms_per_img'])
else: print(open('gpurun_out/r04_bench_p_$c.log').read()[-1500:])
P
done
