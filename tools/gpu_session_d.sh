#!/bin/bash
# GPU session D: GEMM tile-time ablation + race screen, tr-read probe, PMC traffic passes of the bench command.
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 300 python tools/gemm_ablate.py > gpurun_out/r02_gemm_ablate_d.log 2>&1; cat gpurun_out/r02_gemm_ablate_d.log
(/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 tools/probes/tr_read_probe.hip -o /tmp/tr_probe 2>/dev/null && /tmp/tr_probe) > gpurun_out/r02_tr_read_probe.txt 2>&1; head -40 gpurun_out/r02_tr_read_probe.txt
REPO="$GRAFT_REPO_ROOT"
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-parity --no-ref-batch"
timeout 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/pmc_f -o f -- python "$REPO/bench.py" $ARGS > "$REPO/gpurun_out/r02_pmc_f.log" 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/pmc_w -o w -- python "$REPO/bench.py" $ARGS > "$REPO/gpurun_out/r02_pmc_w.log" 2>&1
python "$REPO/tools/pmc_traffic.py" /tmp/pmc_f /tmp/pmc_w "gemm_dma_kernel<256, false, 0>" > "$REPO/gpurun_out/pmc_traffic.json" 2>&1
cat "$REPO/gpurun_out/pmc_traffic.json"
ls /tmp/pmc_f | head; find /tmp/pmc_f -type f | head -5
