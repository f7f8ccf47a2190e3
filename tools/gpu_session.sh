#!/bin/bash
# round 6, session 11: ConvHead prologue form (BatchNorm + GELU in the prediction GEMM's operand load): kernel cases, parity, same-box A/B
cd "$GRAFT_REPO_ROOT" || exit 1
REPO="$GRAFT_REPO_ROOT"; O=$REPO/gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "f32n" > $O/r06_pytest_k_f32n.log 2>&1; echo "f32n ops rc $?"; tail -3 $O/r06_pytest_k_f32n.log
timeout 900 python -m pytest tests/test_gpu_train.py tests/test_gpu_model.py -x -q -m gpu -k "not trajectory and not swin" > $O/r06_pytest_k_train.log 2>&1; echo "train rc $?"; tail -3 $O/r06_pytest_k_train.log
timeout 900 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_configs.py -x -q -m gpu -k "ns6 or cfg2 or cfg3" > $O/r06_pytest_k_full.log 2>&1; echo "fullsize rc $?"; tail -3 $O/r06_pytest_k_full.log
COMMON="--steps 10 --warmup 3 --no-cpu-baseline --no-parity --no-ref-batch --no-torch-baseline --no-fast-mode --no-x3-mode --no-fwd --no-roofline"
for v in "" "--no-head-prologue" "" "--no-head-prologue"; do
  timeout 300 python bench.py $COMMON $v > $O/r06_bench_k_tmp.log 2>&1
  python - "$O/r06_bench_k_tmp.log" "prologue:${v:-on}" <<'PY'
import json, sys
l = [x for x in open(sys.argv[1]) if x.startswith('{')]
if not l:
    print(sys.argv[2], "NO LINE"); print(open(sys.argv[1]).read()[-1500:])
else:
    d = json.loads(l[-1]); print(sys.argv[2], d['value'], d['ms_per_step'], d['peak_hbm_gb'])
PY
done 2>&1 | tee $O/r06_bench_k_head_prologue_ab.log
