#!/bin/bash
# GPU session of the moment (overwritten per session; history in git).  Run as: gpurun --timeout N -- bash tools/gpu_session.sh
# round 6, FINAL-7 (last tree): the whole -m gpu suite, the miniatures on the wide pitch, smoke(), the Swin-B line with all legs
cd "$GRAFT_REPO_ROOT" || exit 1
REPO="$GRAFT_REPO_ROOT"; O=$REPO/gpurun_out; mkdir -p $O
rm -f $O/parity_report.jsonl
timeout 2400 python -m pytest tests/ -q -m gpu > $O/r06_pytest_aq_full.log 2>&1; echo "full suite rc $?"; tail -2 $O/r06_pytest_aq_full.log
MTT_TEST_PITCH32_FROM=33 timeout 1200 python -m pytest tests/test_gpu_model.py tests/test_gpu_train.py tests/test_gpu_ops.py -q -m gpu > $O/r06_pytest_aq_wide_pitch.log 2>&1; echo "wide-pitch suite rc $?"; tail -1 $O/r06_pytest_aq_wide_pitch.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $O/r06_smoke_aq.log 2>&1; echo "smoke rc $?"; tail -1 $O/r06_smoke_aq.log | cut -c1-200
B="--no-torch-baseline --no-cpu-baseline --no-ref-batch --no-x3-mode"
timeout 1500 python bench.py --config swinb --steps 6 --warmup 2 $B > $O/r06_bench_aq_swinb.log 2> $O/r06_bench_aq_swinb.err; echo "swinb rc $?"
python - $O/r06_bench_aq_swinb.log swinb <<'PY'
import json, sys
l=[x for x in open(sys.argv[1]) if x.startswith('{')]
if l:
    d=json.loads(l[-1]); f=d.get('fast_mode') or {}
    print(sys.argv[2], 'batch', d['config']['per_gpu_batch'], {k:d.get(k) for k in ('value','ms_per_step','fwd_ms_per_img','peak_hbm_gb')}, 'bf16', f.get('images_per_s'), f.get('fwd_ms_per_img'), 'parity', (d.get('parity') or {}).get('worst_head_rel_err'))
else: print(sys.argv[2], 'NO LINE', open(sys.argv[1].replace('.log','.err')).read()[-800:])
PY
