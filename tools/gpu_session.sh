#!/bin/bash
# GPU session of the moment (overwritten per session; history in git).  Run as: gpurun --timeout N -- bash tools/gpu_session.sh
# round 6, FINAL-5 (final tree): the whole -m gpu suite, the miniatures on the wide pitch, smoke(), the driver-style bench line + kernel trace,
# the Swin-B line with all legs
cd "$GRAFT_REPO_ROOT" || exit 1
REPO="$GRAFT_REPO_ROOT"; O=$REPO/gpurun_out; mkdir -p $O
rm -f $O/parity_report.jsonl
timeout 2400 python -m pytest tests/ -q -m gpu > $O/r06_pytest_al_full.log 2>&1; echo "full suite rc $?"; tail -2 $O/r06_pytest_al_full.log
MTT_TEST_PITCH32_FROM=33 timeout 1200 python -m pytest tests/test_gpu_model.py tests/test_gpu_train.py tests/test_gpu_ops.py -q -m gpu > $O/r06_pytest_al_wide_pitch.log 2>&1; echo "wide-pitch suite rc $?"; tail -1 $O/r06_pytest_al_wide_pitch.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $O/r06_smoke_al.log 2>&1; echo "smoke rc $?"; tail -1 $O/r06_smoke_al.log | cut -c1-200
timeout 1500 python bench.py --steps 20 --warmup 5 > $O/r06_bench_al_driver_style.log 2> $O/r06_bench_al_driver_style.err; echo "bench rc $?"
python - <<'PY'
import json
l=[x for x in open('gpurun_out/r06_bench_al_driver_style.log') if x.startswith('{')]
if l:
    d=json.loads(l[-1])
    print({k:d[k] for k in ('value','ms_per_step','fwd_ms_per_img','peak_hbm_gb')})
    r=d['roofline']; print({k:v for k,v in r.items() if k in ('achieved','frac','frac_mfma_issued','traffic','traffic_note','launches','kernel_ms_per_step')})
    print('fast', d['fast_mode'] and {k:d['fast_mode'].get(k) for k in ('images_per_s','fwd_ms_per_img','error')}, 'parity', d['parity'] and d['parity'].get('worst_head_rel_err'))
    print('x3', {k: d['full_fp32_mode'].get(k) for k in ('images_per_s','ms_per_step','per_gpu_batch')} if d.get('full_fp32_mode') else None)
    print('roofline_bwd_gemm', (d.get('roofline_bwd_gemm') or {}).get('frac'), 'git', d.get('git'))
PY
cd /tmp; export TMPDIR=/tmp
Q="--no-cpu-baseline --no-torch-baseline --no-ref-batch --no-x3-mode --no-fast-mode --no-parity --no-fwd --no-roofline"
timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_q -o q -- python $REPO/bench.py --steps 3 --warmup 1 $Q > $O/r06_prof_al_run.log 2>&1
python $REPO/tools/prof_summary.py /tmp/prof_q 5 > $O/r06_train_ns6_b126_x3f_al.txt 2>&1
head -5 $O/r06_train_ns6_b126_x3f_al.txt | cut -c1-150
cd $REPO
B="--no-torch-baseline --no-cpu-baseline --no-ref-batch --no-x3-mode"
timeout 1500 python bench.py --config swinb --steps 6 --warmup 2 $B > $O/r06_bench_al_swinb.log 2> $O/r06_bench_al_swinb.err; echo "swinb rc $?"
python - $O/r06_bench_al_swinb.log swinb <<'PY'
import json, sys
l=[x for x in open(sys.argv[1]) if x.startswith('{')]
if l:
    d=json.loads(l[-1]); f=d.get('fast_mode') or {}
    print(sys.argv[2], 'batch', d['config']['per_gpu_batch'], {k:d.get(k) for k in ('value','ms_per_step','fwd_ms_per_img','peak_hbm_gb')}, 'bf16', f.get('images_per_s'), f.get('fwd_ms_per_img'), 'parity', (d.get('parity') or {}).get('worst_head_rel_err'))
else: print(sys.argv[2], 'NO LINE', open(sys.argv[1].replace('.log','.err')).read()[-800:])
PY
