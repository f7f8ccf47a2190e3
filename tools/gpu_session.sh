#!/bin/bash
# GPU session of the moment (overwritten per session; history in git).  Run as: gpurun --timeout N -- bash tools/gpu_session.sh
# round 5: the driver-style bench line on the FINAL tree.
cd "$GRAFT_REPO_ROOT" || exit 1
REPO="$GRAFT_REPO_ROOT"; O=$REPO/gpurun_out; mkdir -p $O
timeout 390 python bench.py --steps 20 --warmup 5 > $O/r05_bench_final.log 2> $O/r05_bench_final.err; echo "bench rc $?"
python - <<'PY'
import json
l=[x for x in open('gpurun_out/r05_bench_final.log') if x.startswith('{')][-1]
d=json.loads(l)
print({k:d[k] for k in ('value','ms_per_step','fwd_ms_per_img','peak_hbm_gb')})
r=d['roofline']; print({k:v for k,v in r.items() if k in ('achieved','frac','frac_mfma_issued','traffic','traffic_note','launches','kernel_ms_per_step')})
print('fast', d['fast_mode']['images_per_s'], d['fast_mode']['fwd_ms_per_img'], 'parity', d['parity']['worst_head_rel_err'], d['full_fp32_mode']['images_per_s'])
PY
