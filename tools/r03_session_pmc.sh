#!/bin/bash
# r03 GPU: SQ counters of the dominant GEMM kernel on the step's shapes (tools/gemm_bench.py): MFMA busy, wave stall buckets, LDS conflicts
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
REPO="$GRAFT_REPO_ROOT"
(cd /tmp && export TMPDIR=/tmp && timeout 200 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE -d /tmp/pmc_sq -o sq -- python "$REPO/tools/gemm_bench.py" --rounds 1 --variant 3 > "$REPO/gpurun_out/r03_pmc_sq_run.log" 2>&1
 python "$REPO/tools/pmc_summary.py" /tmp/pmc_sq gemm_dma_kernel > "$REPO/gpurun_out/r03_pmc_sq_gemm_dma.txt" 2>&1)
tail -3 gpurun_out/r03_pmc_sq_run.log; cat gpurun_out/r03_pmc_sq_gemm_dma.txt | head -40
