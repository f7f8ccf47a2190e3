#!/bin/bash
# GPU session of the moment (overwritten per session; history in git).  Run as: gpurun --timeout N -- bash tools/gpu_session.sh
# round 6, session 1: per-parameter gradient distribution (calibrates tests/train_check.PER_PARAM), the side-stream backward's
# correctness tests, and its same-box A/B on the headline step.
cd "$GRAFT_REPO_ROOT" || exit 1
REPO="$GRAFT_REPO_ROOT"; O=$REPO/gpurun_out; mkdir -p $O
timeout 900 python tools/grad_dist.py --out $O/r06_grad_dist_a.json mini_ctr:x3f mini_ctr:bf16 mini_win:x3f mini_win:bf16 mini_deconv:x3f mini_deconv:bf16 \
    mini_p32:x3f mini8:x3f mini8:bf16 ns6:x3f ns6:bf16 > $O/r06_grad_dist_a.log 2>&1; echo "grad_dist rc $?"
grep -E "^==|VIOLATION" $O/r06_grad_dist_a.log
timeout 900 python -m pytest tests/test_gpu_train.py -x -q -m gpu -k "not trajectory and not swin" > $O/r06_pytest_a_train.log 2>&1; echo "pytest rc $?"; tail -3 $O/r06_pytest_a_train.log
COMMON="--steps 10 --warmup 3 --no-cpu-baseline --no-parity --no-ref-batch --no-torch-baseline --no-fast-mode --no-x3-mode --no-fwd"
for v in "1 0" "0 0" "1 -1" "0 0" "1 0"; do
  set -- $v
  timeout 300 python bench.py $COMMON --side-stream $1 --side-priority $2 > $O/r06_bench_a_side$1_p$2.log 2>&1
  python - "$O/r06_bench_a_side$1_p$2.log" "$v" <<'PY'
import json, sys
l = [x for x in open(sys.argv[1]) if x.startswith('{')]
if not l:
    print("side", sys.argv[2], "NO LINE"); print(open(sys.argv[1]).read()[-1500:])
else:
    d = json.loads(l[-1]); print("side/prio", sys.argv[2], d['value'], d['ms_per_step'])
PY
done
