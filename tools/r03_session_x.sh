#!/bin/bash
# r03 GPU session X: the other BASELINE configs and Swin-B on the final code (short bench lines)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
S="--steps 4 --warmup 2 --no-cpu-baseline --no-roofline --no-parity --no-parity-mode --no-ref-batch --no-torch-baseline"
for c in cfg2 cfg3 cfg4 cfg5 swinb; do
  timeout 400 python bench.py --config $c $S > gpurun_out/r03_bench_x_$c.log 2>&1
  python - "$c" <<'PY'
import json, sys
c = sys.argv[1]
for l in open(f'gpurun_out/r03_bench_x_{c}.log'):
    if l.startswith('{"metric"'):
        d = json.loads(l); print(c, 'VALUE', d['value'], 'img/s', d['ms_per_step'], 'ms/step  fwd', d['fwd_ms_per_img'], 'batch', d['config']['per_gpu_batch'], 'hbm', d['peak_hbm_gb'])
PY
done
