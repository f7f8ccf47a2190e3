#!/bin/bash
# GPU session M: persistent LDS-DMA GEMM kernel (gemm_pdma_kernel): op parity vs the emulator, race screen + micro-benchmark A/B,
# model / training-step parity with the persistent kernel forced (MTT_TEST_GEMM_VARIANT=12), whole-step A/B.
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_gpu_ops.py -m gpu -q -rf -k "pdma" > gpurun_out/r02_pytest_m_ops.log 2>&1
tail -8 gpurun_out/r02_pytest_m_ops.log
timeout 300 python tools/gemm_bench.py 5 > gpurun_out/r02_gemm_bench_m.log 2>&1
tail -12 gpurun_out/r02_gemm_bench_m.log
for v in 12 13; do
  MTT_TEST_GEMM_VARIANT=$v timeout 200 python -m pytest tests/test_gpu_fullsize.py -m gpu -q -rf -k "dma_gemm_and_conv" > gpurun_out/r02_pytest_m_benchshapes_v$v.log 2>&1
  tail -3 gpurun_out/r02_pytest_m_benchshapes_v$v.log
done
for v in 14 12 13; do
  timeout 300 python bench.py --steps 6 --warmup 2 --gemm-variant $v --no-cpu-baseline --no-parity --no-ref-batch 2>&1 | tail -1 | cut -c1-2400 > gpurun_out/r02_bench_m_v$v.log
  python - <<PY
import json
r = json.loads(open("gpurun_out/r02_bench_m_v$v.log").read())
print("variant $v:", r["value"], "img/s", r["ms_per_step"], "ms; fwd", r["fwd_ms_per_img"], "ms/img; roofline", r["roofline"]["achieved"], r["roofline"]["frac"], "launches", r["roofline"]["launches"], "persistent", r["roofline"].get("persistent_launches"))
PY
done
