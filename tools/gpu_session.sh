#!/bin/bash
# GPU session of the moment (overwritten per session; history in git).  Run as: gpurun --timeout N -- bash tools/gpu_session.sh
# round 5 evidence on the final sources: SQ counters (MFMA pipe busy, waits, LDS conflicts) of the GEMM and attention kernels, and the
# kernel-trace summary of the x3f inference forward at the benchmark's batch.
cd "$GRAFT_REPO_ROOT" || exit 1
REPO="$GRAFT_REPO_ROOT"; O=$REPO/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
C="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE"
timeout 200 rocprofv3 --kernel-trace --pmc $C -d /tmp/pmc_sq -o sq -- python $REPO/tools/gemm_bench.py --split --rounds 1 --no-check > $O/r05_pmc_sq_run.log 2>&1
python $REPO/tools/pmc_summary.py /tmp/pmc_sq gemm_ring3 gemm_dma_kernel > $O/r05_pmc_sq_gemm.txt 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc $C -d /tmp/pmc_sq2 -o sq2 -- python $REPO/tools/dec_x3_bench.py > $O/r05_pmc_sq_dec_run.log 2>&1
python $REPO/tools/pmc_summary.py /tmp/pmc_sq2 gemm_ring3 > $O/r05_pmc_sq_gemm_decoder_shapes.txt 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc $C -d /tmp/pmc_sq3 -o sq3 -- python $REPO/tools/attn_x3_bench.py 63 1030 16 6 > $O/r05_pmc_sq_attn_run.log 2>&1
python $REPO/tools/pmc_summary.py /tmp/pmc_sq3 attn_fwd > $O/r05_pmc_sq_attn.txt 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_fwd -o fwd -- python $REPO/tools/fwd_bench.py --batch 63 --iters 4 --warmup 1 --prec x3f > $O/r05_prof_fwd_run.log 2>&1
python $REPO/tools/prof_summary.py /tmp/prof_fwd 5 > $O/r05_fwd_x3f_b63.txt 2>&1
grep "fwd TaskPrompter" $O/r05_prof_fwd_run.log; head -14 $O/r05_fwd_x3f_b63.txt | cut -c1-130; cat $O/r05_pmc_sq_gemm.txt | head -24; cat $O/r05_pmc_sq_attn.txt | head -24
