#!/bin/bash
# r03 GPU: SQ counters of the flash-attention kernels (tools/attn_bench.py): MFMA busy vs VALU activity
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
REPO="$GRAFT_REPO_ROOT"
(cd /tmp && export TMPDIR=/tmp && timeout 150 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE -d /tmp/pmc_at -o at -- python "$REPO/tools/attn_bench.py" 63 1030 16 6 > "$REPO/gpurun_out/r03_pmc_attn_run.log" 2>&1
 python "$REPO/tools/pmc_summary.py" /tmp/pmc_at attn_ > "$REPO/gpurun_out/r03_pmc_sq_attn.txt" 2>&1)
tail -4 gpurun_out/r03_pmc_attn_run.log | cut -c1-200; cat gpurun_out/r03_pmc_sq_attn.txt | head -60
