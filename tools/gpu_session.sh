#!/bin/bash
# GPU session of the moment (overwritten per session; history in git).  Run as: gpurun --timeout N -- bash tools/gpu_session.sh
# round 5, last session: the M-edge form of the weight-gradient kernel (tests, same-box A/B of ALL edge forms against a build without them),
# then the PMC traffic passes, the driver-style bench line and the kernel-trace summary on the final gemm.hip.
cd "$GRAFT_REPO_ROOT" || exit 1
REPO="$GRAFT_REPO_ROOT"; O=$REPO/gpurun_out; mkdir -p $O
rm -f $O/pmc_traffic.json
timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -k "gemm_tn or conv3 or gemm_split" > $O/r05_pytest_n_ops.log 2>&1; tail -3 $O/r05_pytest_n_ops.log
timeout 900 python -m pytest tests/test_gpu_train.py tests/test_gpu_fullsize.py -x -q -k "not swin and not trajectory" > $O/r05_pytest_n_train.log 2>&1; tail -3 $O/r05_pytest_n_train.log
Q="--steps 10 --warmup 3 --no-cpu-baseline --no-torch-baseline --no-ref-batch --no-x3-mode --no-fast-mode --no-parity --no-fwd --no-roofline"
for mode in edge noedge edge noedge; do
python - $mode $Q <<'PY' 2>/dev/null | python -c "import sys, json; [print(sys.argv[1], json.loads(l)['value'], json.loads(l)['ms_per_step']) for l in sys.stdin if l.startswith('{')]" $mode | tee -a $O/r05_edge_ab_n.log
import sys, os
sys.path.insert(0, '.')
mode = sys.argv[1]
sys.argv = ['bench.py'] + sys.argv[2:]
import mtt_amd
if mode == 'noedge':
    mtt_amd._lib.LIB_PATH = os.path.abspath('build/variants/libmtt_noedge.so')
import bench
bench.main()
PY
done
cd /tmp && export TMPDIR=/tmp
Q2="--steps 1 --warmup 1 --no-cpu-baseline --no-torch-baseline --no-roofline --no-parity --no-ref-batch --no-fast-mode --no-x3-mode --no-fwd"
timeout 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/pmc_f -o f -- python $REPO/bench.py $Q2 > $O/r05_pmc_fetch_run.log 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/pmc_w -o w -- python $REPO/bench.py $Q2 > $O/r05_pmc_write_run.log 2>&1
cd $REPO
python tools/pmc_traffic.py /tmp/pmc_f /tmp/pmc_w 'gemm_ring3_kernel' $O/pmc_traffic.json > /dev/null
python tools/pmc_traffic.py /tmp/pmc_f /tmp/pmc_w 'gemm_dma_kernel<1>' $O/pmc_traffic.json > /dev/null
cp $O/pmc_traffic.json profiles/pmc_traffic.json
timeout 600 python bench.py --steps 20 --warmup 5 > $O/r05_bench_zzz.log 2> $O/r05_bench_zzz.err; echo "bench rc $?"
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_zzz -o zzz -- python $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-torch-baseline --no-ref-batch --no-x3-mode --no-fast-mode --no-parity --no-fwd --no-roofline > $O/r05_prof_zzz_run.log 2>&1
python $REPO/tools/prof_summary.py /tmp/prof_zzz 5 > $O/r05_train_ns6_b63_x3f_zzz.txt 2>&1
cd $REPO
python - <<'PY'
import json
l=[x for x in open('gpurun_out/r05_bench_zzz.log') if x.startswith('{')][-1]
d=json.loads(l)
print({k:d[k] for k in ('value','ms_per_step','fwd_ms_per_img','peak_hbm_gb')})
r=d['roofline']; print({k:v for k,v in r.items() if k in ('achieved','frac','frac_mfma_issued','traffic','traffic_note','launches','kernel_ms_per_step')})
print('fast', d['fast_mode']['images_per_s'], d['fast_mode']['fwd_ms_per_img'], 'parity', d['parity']['worst_head_rel_err'])
PY
