#!/bin/bash
# round 6, FINAL-2d: Swin-B after the WinAttnFn reference-cycle fix: steady memory? batch 8 / 12 / 16 bare loops (8 steps), then the full line at 16
cd "$GRAFT_REPO_ROOT" || exit 1
REPO="$GRAFT_REPO_ROOT"; O=$REPO/gpurun_out; mkdir -p $O
B="--no-torch-baseline --no-cpu-baseline --no-ref-batch --no-x3-mode"
show() { python - $1 "$2" <<'PY'
import json, sys
l=[x for x in open(sys.argv[1]) if x.startswith('{')]
if l:
    d=json.loads(l[-1]); f=d.get('fast_mode') or {}
    print(sys.argv[2], d['config']['per_gpu_batch'], {k:d.get(k) for k in ('value','ms_per_step','fwd_ms_per_img','peak_hbm_gb')}, 'bf16', f.get('images_per_s'), 'parity', (d.get('parity') or {}).get('worst_head_rel_err'))
else: print(sys.argv[2], 'NO LINE', open(sys.argv[1].replace('.log','.err')).read()[-600:])
PY
}
timeout 600 python -m pytest tests -m gpu -q -k "swin or Swin" 2>&1 | tail -2
for b in 8 12 16; do
  timeout 900 python bench.py --config swinb --batch $b --steps 6 --warmup 2 $B --no-fast-mode --no-parity --no-roofline > $O/t$b.log 2>$O/t$b.err; show $O/t$b.log "bare s6w2"
done
timeout 1200 python bench.py --config swinb --batch 16 --steps 6 --warmup 2 $B > $O/r06_bench_af_swinb_b16.log 2>$O/r06_bench_af_swinb_b16.err; show $O/r06_bench_af_swinb_b16.log "full b16"
timeout 1200 python bench.py --config swinb --batch 8 --steps 6 --warmup 2 $B > $O/r06_bench_af_swinb_b8.log 2>$O/r06_bench_af_swinb_b8.err; show $O/r06_bench_af_swinb_b8.log "full b8"
