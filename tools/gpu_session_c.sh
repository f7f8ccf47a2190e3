#!/bin/bash
# GPU session C: balanced LDS-DMA schedule A/B, permlane probe, the other BASELINE configs through bench.py, NS-6 bench line.
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_gpu_ops.py -m gpu -q -rf -k "gemm or bn_ or attn_msg" > gpurun_out/r02_pytest_c_ops.log 2>&1
tail -6 gpurun_out/r02_pytest_c_ops.log
timeout 300 python tools/gemm_bench.py 5 > gpurun_out/r02_gemm_bench_c.log 2>&1; cat gpurun_out/r02_gemm_bench_c.log
(/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 tools/probes/permlane_probe.hip -o /tmp/permlane_probe && /tmp/permlane_probe) > gpurun_out/r02_permlane_probe.txt 2>&1; head -12 gpurun_out/r02_permlane_probe.txt
for c in cfg4 cfg2 cfg3 cfg5; do
  timeout 420 python bench.py --config $c --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r02_bench_c_$c.log 2>&1
  tail -c 1800 gpurun_out/r02_bench_c_$c.log; echo
done
timeout 400 python bench.py --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/r02_bench_c.log 2>&1
tail -c 1500 gpurun_out/r02_bench_c.log
