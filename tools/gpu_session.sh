#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
REPO="$GRAFT_REPO_ROOT"; O=$REPO/gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py tests/test_gpu_train.py -m gpu -x -q > $O/r04_pytest_i.log 2>&1; tail -5 $O/r04_pytest_i.log
timeout 900 python bench.py --steps 8 --warmup 3 --no-torch-baseline --no-ref-batch > $O/r04_bench_i.log 2>&1
python - <<'P'
import json
l=[x for x in open('gpurun_out/r04_bench_i.log') if x.startswith('{')]
if l:
    d=json.loads(l[-1]); print('x3f', d['value'], d['ms_per_step'], d['fwd_ms_per_img'], d['parity']); print('bf16', d['fast_mode']['images_per_s'], d['fast_mode']['fwd_ms_per_img'])
else: print(open('gpurun_out/r04_bench_i.log').read()[-2000:])
P
S="--steps 4 --warmup 2 --no-torch-baseline --no-ref-batch --no-fast-mode --no-cpu-baseline --no-roofline"
timeout 600 python bench.py --config cfg4 $S > $O/r04_bench_i_cfg4.log 2>&1; tail -c 400 $O/r04_bench_i_cfg4.log | head -c 300; python -c "
import json; d=json.loads([x for x in open('gpurun_out/r04_bench_i_cfg4.log') if x.startswith('{')][-1]); print('cfg4 x3f', d['value'], d['fwd_ms_per_img'])"
