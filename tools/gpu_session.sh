#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
REPO="$GRAFT_REPO_ROOT"; O=$REPO/gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "conv3_split or gemm_split or modulate" > $O/r04_pytest_o1.log 2>&1; tail -5 $O/r04_pytest_o1.log
timeout 1500 python -m pytest tests/test_gpu_configs.py tests/test_gpu_fullsize.py tests/test_gpu_train.py -m gpu -x -q -k "not swinb" > $O/r04_pytest_o2.log 2>&1; tail -5 $O/r04_pytest_o2.log
timeout 900 python bench.py --steps 8 --warmup 3 --no-torch-baseline --no-ref-batch --no-x3-mode > $O/r04_bench_o.log 2>&1
python - <<'P'
import json
l=[x for x in open('gpurun_out/r04_bench_o.log') if x.startswith('{')]
if l:
    d=json.loads(l[-1]); print('x3f', d['value'], d['ms_per_step'], d['fwd_ms_per_img'], d['parity']['worst_head_rel_err'], d['roofline']['frac']); print('bf16', d['fast_mode']['images_per_s'], d['fast_mode']['fwd_ms_per_img'])
    for s in d['roofline']['by_shape']: print('   ', s)
else: print(open('gpurun_out/r04_bench_o.log').read()[-2000:])
P
