#!/bin/bash
# GPU session P: TaskPrompter-Swin forward kernels (op parity vs the emulator, miniature models vs the reference golden, Swin-B at full
# size vs the CPU oracle, forward throughput) + the fast source addressing of the 256 x 256 LDS-DMA GEMM (every gemm op case, step A/B).
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
rm -f gpurun_out/parity_report.jsonl
timeout 300 python -m pytest tests/test_gpu_ops.py -m gpu -q -rf -k "patchify_P or resize_nchw or gather_ or winattn or chanattn or conv3s2 or modulate_hg32" > gpurun_out/r02_pytest_p_swin_ops.log 2>&1
tail -12 gpurun_out/r02_pytest_p_swin_ops.log
timeout 300 python -m pytest tests/test_gpu_model.py -m gpu -q -rf -s -k "swin" > gpurun_out/r02_pytest_p_swin_model.log 2>&1
grep -h "PARITY\|passed\|failed" gpurun_out/r02_pytest_p_swin_model.log | tail -14
timeout 400 python -m pytest tests/test_gpu_configs.py -m gpu -q -rf -k "cs_swinB" > gpurun_out/r02_pytest_p_swin_full.log 2>&1
tail -6 gpurun_out/r02_pytest_p_swin_full.log
timeout 300 python tools/swin_bench.py bf16 1,4 > gpurun_out/r02_swin_bench_p.log 2>&1
tail -3 gpurun_out/r02_swin_bench_p.log
timeout 400 python -m pytest tests/test_gpu_ops.py -m gpu -q -rf -k "gemm" > gpurun_out/r02_pytest_p_gemm_ops.log 2>&1
tail -4 gpurun_out/r02_pytest_p_gemm_ops.log
for v in 18 0; do
  timeout 300 python bench.py --steps 6 --warmup 2 --gemm-variant $v --no-cpu-baseline --no-parity --no-ref-batch 2>&1 | tail -1 | cut -c1-2600 > gpurun_out/r02_bench_p_v$v.log
  python - <<PY
import json
r = json.loads(open("gpurun_out/r02_bench_p_v$v.log").read())
print("variant $v:", r["value"], "img/s", r["ms_per_step"], "ms; fwd", r["fwd_ms_per_img"], "ms/img; roofline", r["roofline"]["achieved"], r["roofline"]["frac"], "launches", r["roofline"]["launches"], "loss", r["config"]["loss"])
PY
done
cp gpurun_out/parity_report.jsonl gpurun_out/r02_parity_report_p.jsonl 2>/dev/null
