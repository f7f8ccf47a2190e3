#!/bin/bash
# GPU session of the moment (overwritten per session; history in git).  Run as: gpurun --timeout N -- bash tools/gpu_session.sh
# FINAL session of round 5: PMC traffic of the dominant kernels on the final gemm.hip, the whole -m gpu suite, the driver-style bench line,
# the kernel-trace summary of the training step.
cd "$GRAFT_REPO_ROOT" || exit 1
REPO="$GRAFT_REPO_ROOT"; O=$REPO/gpurun_out; mkdir -p $O
rm -f $O/parity_report.jsonl $O/pmc_traffic.json
cd /tmp && export TMPDIR=/tmp
Q="--steps 1 --warmup 1 --no-cpu-baseline --no-torch-baseline --no-roofline --no-parity --no-ref-batch --no-fast-mode --no-x3-mode --no-fwd"
timeout 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/pmc_f -o f -- python $REPO/bench.py $Q > $O/r05_pmc_fetch_run.log 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/pmc_w -o w -- python $REPO/bench.py $Q > $O/r05_pmc_write_run.log 2>&1
cd $REPO
python tools/pmc_traffic.py /tmp/pmc_f /tmp/pmc_w 'gemm_ring3_kernel' $O/pmc_traffic.json
python tools/pmc_traffic.py /tmp/pmc_f /tmp/pmc_w 'gemm_dma_kernel<1>' $O/pmc_traffic.json
cp $O/pmc_traffic.json profiles/pmc_traffic.json
timeout 900 python -m pytest tests -m gpu -x -q > $O/r05_pytest_zz.log 2>&1; echo "pytest rc $?" >> $O/r05_pytest_zz.log
tail -5 $O/r05_pytest_zz.log
cp $O/parity_report.jsonl $O/r05_parity_report_zz.jsonl 2>/dev/null
timeout 600 python bench.py --steps 20 --warmup 5 > $O/r05_bench_zz.log 2> $O/r05_bench_zz.err; echo "bench rc $?"
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_zz -o zz -- python $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-torch-baseline --no-ref-batch --no-x3-mode --no-fast-mode --no-parity --no-fwd --no-roofline > $O/r05_prof_zz_run.log 2>&1
python $REPO/tools/prof_summary.py /tmp/prof_zz 5 > $O/r05_train_ns6_b63_x3f_zz.txt 2>&1
cd $REPO
python - <<'PY'
import json
l=[x for x in open('gpurun_out/r05_bench_zz.log') if x.startswith('{')][-1]
d=json.loads(l)
print({k:d[k] for k in ('value','ms_per_step','fwd_ms_per_img','peak_hbm_gb','host')})
r=d['roofline']; print({k:v for k,v in r.items() if k not in ('by_shape','kernel','traffic_source','flop_convention')})
print('bwd', {k:v for k,v in d['roofline_bwd_gemm'].items() if k in ('achieved','frac','traffic','kernel_ms_per_step','launches','traffic_note')})
f=d['fast_mode']; print('fast', f['images_per_s'], f['fwd_ms_per_img'], f['roofline']['frac'], f['parity']['worst_head_rel_err'])
print('parity', d['parity']['worst_head_rel_err'], 'full', d['full_fp32_mode']['images_per_s'], 'ref_batch', d['ref_batch'], d['git'])
PY
