#!/bin/bash
# round 6, session 31: ConvTranspose2d as a split-plane GEMM + mtt_pixshuf2 (x3f): op parity, Swin / cfg5 model parity, A/B benches
cd "$GRAFT_REPO_ROOT" || exit 1
REPO="$GRAFT_REPO_ROOT"; O=$REPO/gpurun_out; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -q -k "pixshuf" 2>&1 | tail -3
timeout 1500 python -m pytest tests -m gpu -q -k "swin or Swin or deconv or cfg5" 2>&1 | tail -4
B="--no-torch-baseline --no-cpu-baseline --no-ref-batch --no-x3-mode --no-fast-mode --no-parity --no-roofline"
show() { python - $1 "$2" <<'PY'
import json, sys
l=[x for x in open(sys.argv[1]) if x.startswith('{')]
if l:
    d=json.loads(l[-1]); print(sys.argv[2], d['config']['per_gpu_batch'], {k:d.get(k) for k in ('value','ms_per_step','fwd_ms_per_img','peak_hbm_gb')})
else: print(sys.argv[2], 'NO LINE', open(sys.argv[1].replace('.log','.err')).read()[-600:])
PY
}
for rep in 1 2; do
for ds in 1 0; do
  MTT_DECONV_SPLIT=$ds timeout 900 python bench.py --config swinb --steps 5 --warmup 2 $B > $O/r06_bench_ah_swinb_ds$ds.log 2>$O/r06_bench_ah_swinb_ds$ds.err; show $O/r06_bench_ah_swinb_ds$ds.log "swinb deconv_split=$ds"
done
done
for ds in 1 0; do
  MTT_DECONV_SPLIT=$ds timeout 900 python bench.py --config cfg5 --steps 3 --warmup 1 $B > $O/r06_bench_ah_cfg5_ds$ds.log 2>$O/r06_bench_ah_cfg5_ds$ds.err; show $O/r06_bench_ah_cfg5_ds$ds.log "cfg5 deconv_split=$ds"
done
