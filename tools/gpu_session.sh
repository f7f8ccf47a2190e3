#!/bin/bash
# GPU session of the moment (overwritten per session; history in git).  Run as: gpurun --timeout N -- bash tools/gpu_session.sh
# round 6, FINAL-1: the whole -m gpu suite + smoke(), the driver-style bench line, the kernel trace of the same step, the PMC traffic passes
# (re-measured on the final csrc/gemm.hip) and the SQ counters of the GEMM / attention kernels.
cd "$GRAFT_REPO_ROOT" || exit 1
REPO="$GRAFT_REPO_ROOT"; O=$REPO/gpurun_out; mkdir -p $O
rm -f $O/parity_report.jsonl $O/pmc_traffic.json
timeout 2400 python -m pytest tests/ -q -m gpu > $O/r06_pytest_q_full.log 2>&1; echo "full suite rc $?"; tail -3 $O/r06_pytest_q_full.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $O/r06_smoke_q.log 2>&1; echo "smoke rc $?"; tail -2 $O/r06_smoke_q.log
timeout 1500 python bench.py --steps 20 --warmup 5 > $O/r06_bench_q_driver_style.log 2> $O/r06_bench_q_driver_style.err; echo "bench rc $?"
python - <<'PY'
import json
l=[x for x in open('gpurun_out/r06_bench_q_driver_style.log') if x.startswith('{')]
if l:
    d=json.loads(l[-1])
    print({k:d[k] for k in ('value','ms_per_step','fwd_ms_per_img','peak_hbm_gb')})
    r=d['roofline']; print({k:v for k,v in r.items() if k in ('achieved','frac','frac_mfma_issued','traffic','traffic_note','launches','kernel_ms_per_step')})
    print('fast', d['fast_mode'] and {k:d['fast_mode'].get(k) for k in ('images_per_s','fwd_ms_per_img','error')}, 'parity', d['parity'] and d['parity'].get('worst_head_rel_err'))
    print('x3', d['full_fp32_mode'])
    print('ref_batch', d['ref_batch'])
PY
cd /tmp; export TMPDIR=/tmp
Q="--no-cpu-baseline --no-torch-baseline --no-ref-batch --no-x3-mode --no-fast-mode --no-parity --no-fwd --no-roofline"
timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_q -o q -- python $REPO/bench.py --steps 3 --warmup 1 $Q > $O/r06_prof_q_run.log 2>&1
python $REPO/tools/prof_summary.py /tmp/prof_q 5 > $O/r06_train_ns6_b126_x3f_final.txt 2>&1
head -4 $O/r06_train_ns6_b126_x3f_final.txt | cut -c1-150
timeout 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/pmc_f -o f -- python $REPO/bench.py --steps 1 --warmup 1 $Q > $O/r06_pmc_f_run.log 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/pmc_w -o w -- python $REPO/bench.py --steps 1 --warmup 1 $Q > $O/r06_pmc_w_run.log 2>&1
cd $REPO
python tools/pmc_traffic.py /tmp/pmc_f /tmp/pmc_w 'gemm_ring3_kernel' $O/pmc_traffic.json
python tools/pmc_traffic.py /tmp/pmc_f /tmp/pmc_w 'gemm_dma_kernel<1>' $O/pmc_traffic.json
cat $O/pmc_traffic.json | head -40
