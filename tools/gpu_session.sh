#!/bin/bash
# GPU session of the moment (overwritten per session; history in git).  Run as: gpurun --timeout N -- bash tools/gpu_session.sh
cd "$GRAFT_REPO_ROOT" || exit 1
REPO="$GRAFT_REPO_ROOT"; O=$REPO/gpurun_out; mkdir -p $O
export MTT_COMMIT=ab78429
B="--no-cpu-baseline --no-roofline --no-parity --no-fast-mode --no-x3-mode --no-ref-batch --no-torch-baseline --no-fwd"
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/pmc_f -o f -- python $REPO/bench.py --prec x3f --steps 1 --warmup 1 $B > $O/r04_pmc_f.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/pmc_w -o w -- python $REPO/bench.py --prec x3f --steps 1 --warmup 1 $B > $O/r04_pmc_w.log 2>&1
cd $REPO
python tools/pmc_traffic.py /tmp/pmc_f /tmp/pmc_w 'gemm_ring3_kernel' profiles/pmc_traffic.json
python tools/pmc_traffic.py /tmp/pmc_f /tmp/pmc_w 'gemm_dma_kernel<1>' profiles/pmc_traffic.json
cp profiles/pmc_traffic.json $O/pmc_traffic.json
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_n -o train -- python $REPO/bench.py --prec x3f --steps 2 --warmup 1 $B > $O/r04_prof_n.log 2>&1
python $REPO/tools/prof_summary.py /tmp/prof_n 4 > $O/r04_train_ns6_b63_x3f_final.txt 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_nf -o fwd -- python $REPO/tools/fwd_bench.py --batch 63 --iters 3 --warmup 1 --prec x3f > $O/r04_prof_nf.log 2>&1
python $REPO/tools/prof_summary.py /tmp/prof_nf 4 > $O/r04_fwd_x3f_b63_final.txt 2>&1
cd $REPO
timeout 1200 python bench.py --steps 20 --warmup 5 > $O/r04_bench_final_driver_style.log 2>&1
tail -c 300 $O/r04_bench_final_driver_style.log
