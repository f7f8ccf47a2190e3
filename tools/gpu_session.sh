#!/bin/bash
# GPU session of the moment (overwritten per session; history in git).  Run as: gpurun --timeout N -- bash tools/gpu_session.sh
cd "$GRAFT_REPO_ROOT" || exit 1
REPO="$GRAFT_REPO_ROOT"; O=$REPO/gpurun_out; mkdir -p $O
timeout 420 python bench.py --config cfg4 --steps 8 --warmup 2 --no-cpu-baseline --no-torch-baseline --no-ref-batch --no-x3-mode > $O/r05_bench_o_cfg4.log 2> $O/r05_bench_o_cfg4.err; echo "cfg4 rc $?"
python - <<'PY'
import json
l = [x for x in open('gpurun_out/r05_bench_o_cfg4.log') if x.startswith('{')][-1]
d = json.loads(l)
f = d.get('fast_mode') or {}
print('cfg4 x3f', d['value'], 'img/s', d['ms_per_step'], 'ms fwd', d['fwd_ms_per_img'], 'parity', (d.get('parity') or {}).get('worst_head_rel_err'), '| bf16', f.get('images_per_s'), f.get('fwd_ms_per_img'))
PY
