#!/bin/bash
# r03 GPU session FINAL: FULL -m gpu suite, smoke, driver-style bench line, x3f main line, step profile (launch count) on the final code
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
REPO="$GRAFT_REPO_ROOT"
rm -f gpurun_out/parity_report.jsonl
timeout 1500 python -m pytest tests -m gpu -q -rf --durations=5 > gpurun_out/r03_pytest_final_full.log 2>&1
tail -14 gpurun_out/r03_pytest_final_full.log
cp gpurun_out/parity_report.jsonl gpurun_out/r03_parity_report_final_full_suite.jsonl 2>/dev/null
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r03_smoke_final.log 2>&1; tail -2 gpurun_out/r03_smoke_final.log
SECONDS=0
timeout 900 python bench.py > gpurun_out/r03_bench_final_driver_style.log 2>&1
echo "bench.py (default flags) took ${SECONDS}s"
python - <<'PY'
import json
for l in open('gpurun_out/r03_bench_final_driver_style.log'):
    if l.startswith('{"metric"'):
        d=json.loads(l)
        print('VALUE', d['value'], d['ms_per_step'], 'fwd', d['fwd_ms_per_img'], 'host', d['host'])
        print('roofline', {k: d['roofline'][k] for k in ('achieved','frac','traffic','launches','kernel_ms_per_step')})
        print('parity', d['parity']['worst_head_rel_err'], 'parity_mode', d['parity_mode']['images_per_s'], d['parity_mode']['worst_head_rel_err'])
        print('ref_batch', d['ref_batch']['images_per_s'], d['ref_batch']['graphed']['images_per_s'])
        print('torch', d['torch_rocm_baseline']['fp32'], d['torch_rocm_baseline']['bf16'])
        print('cpu', d['cpu_baseline']['value'])
PY
B="--no-cpu-baseline --no-roofline --no-parity --no-parity-mode --no-ref-batch --no-torch-baseline"
