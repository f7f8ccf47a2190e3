#!/bin/bash
# r03 GPU session R: register-resident LayerNorm forward, modulate_bwd templated on the task count: parity + per-kernel times from a step profile
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
REPO="$GRAFT_REPO_ROOT"
timeout 300 python -m pytest tests/test_gpu_ops.py -m gpu -q -rf -k "ln_ or layernorm or colsum or bn_" > gpurun_out/r03_pytest_r_ops.log 2>&1; tail -4 gpurun_out/r03_pytest_r_ops.log
B="--no-cpu-baseline --no-roofline --no-parity --no-parity-mode --no-ref-batch --no-torch-baseline"
(cd /tmp && export TMPDIR=/tmp && timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_r -o train -- python "$REPO/bench.py" --steps 3 --warmup 1 --no-fwd $B > "$REPO/gpurun_out/r03_prof_r.log" 2>&1; python "$REPO/tools/prof_summary.py" /tmp/prof_r 5 > "$REPO/gpurun_out/r03_train_ns6_b63_r.txt" 2>&1)
grep -n "dispatches\|ln_dgb\|colsum_final\|ln_bwd\|gemm_dma_kernel<1>" gpurun_out/r03_train_ns6_b63_r.txt | head -12 | cut -c1-170
