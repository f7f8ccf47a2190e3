#!/bin/bash
# GPU session of the moment (overwritten per session; history in git).  Run as: gpurun --timeout N -- bash tools/gpu_session.sh
cd "$GRAFT_REPO_ROOT" || exit 1
REPO="$GRAFT_REPO_ROOT"; O=$REPO/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE -d /tmp/pmc_sq -o sq -- python $REPO/tools/gemm_bench.py --split --rounds 1 --no-check > $O/r04_pmc_sq_run.log 2>&1
python $REPO/tools/pmc_summary.py /tmp/pmc_sq gemm_ring3 gemm_dma_kernel > $O/r04_pmc_sq_gemm.txt 2>&1
cat $O/r04_pmc_sq_gemm.txt | head -40
