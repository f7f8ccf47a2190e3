#!/bin/bash
# r03 GPU session M: FULL -m gpu suite, smoke, the driver-style bench line with every leg, PMC traffic of the dominant kernel, the other
# BASELINE configs (short bench lines).
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
REPO="$GRAFT_REPO_ROOT"
rm -f gpurun_out/parity_report.jsonl
timeout 1500 python -m pytest tests -m gpu -q -rf --durations=8 > gpurun_out/r03_pytest_m_full.log 2>&1
tail -22 gpurun_out/r03_pytest_m_full.log
cp gpurun_out/parity_report.jsonl gpurun_out/r03_parity_report_m_full_suite.jsonl 2>/dev/null
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r03_smoke_m.log 2>&1; tail -2 gpurun_out/r03_smoke_m.log
SECONDS=0
timeout 900 python bench.py > gpurun_out/r03_bench_m_driver_style.log 2>&1
echo "bench.py (default flags) took ${SECONDS}s"
tail -c 5000 gpurun_out/r03_bench_m_driver_style.log; echo
B="--steps 1 --warmup 1 --no-fwd --no-cpu-baseline --no-roofline --no-parity --no-parity-mode --no-ref-batch --no-torch-baseline"
(cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/pmc_f -o f -- python "$REPO/bench.py" $B > /dev/null 2>&1
 timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/pmc_w -o w -- python "$REPO/bench.py" $B > /dev/null 2>&1
 python "$REPO/tools/pmc_traffic.py" /tmp/pmc_f /tmp/pmc_w 'gemm_dma_kernel<1>' > "$REPO/gpurun_out/r03_pmc_traffic_m.json" 2>&1)
cat gpurun_out/r03_pmc_traffic_m.json
S="--steps 4 --warmup 2 --no-cpu-baseline --no-roofline --no-parity --no-parity-mode --no-ref-batch --no-torch-baseline"
for c in cfg2 cfg3 cfg4 cfg5; do
  timeout 400 python bench.py --config $c $S > gpurun_out/r03_bench_m_$c.log 2>&1
  python - "$c" <<'PY'
import json, sys
c = sys.argv[1]
for l in open(f'gpurun_out/r03_bench_m_{c}.log'):
    if l.startswith('{"metric"'):
        d = json.loads(l); print(c, 'VALUE', d['value'], 'img/s', d['ms_per_step'], 'ms/step  fwd', d['fwd_ms_per_img'], 'batch', d['config']['per_gpu_batch'], 'hbm', d['peak_hbm_gb'])
PY
done
