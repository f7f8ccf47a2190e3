#!/bin/bash
# round 6, session 10: dead side-channel work removed (no materialised zero gradients, channel logits only at the taps), one DropPath
# draw per step: training parity + same-box bench + kernel trace
cd "$GRAFT_REPO_ROOT" || exit 1
REPO="$GRAFT_REPO_ROOT"; O=$REPO/gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_train.py tests/test_gpu_model.py -x -q -m gpu -k "not trajectory and not swin" > $O/r06_pytest_j_train.log 2>&1; echo "train rc $?"; tail -3 $O/r06_pytest_j_train.log
timeout 900 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_configs.py -x -q -m gpu -k "ns6 or cfg4" > $O/r06_pytest_j_full.log 2>&1; echo "fullsize rc $?"; tail -3 $O/r06_pytest_j_full.log
COMMON="--steps 10 --warmup 3 --no-cpu-baseline --no-parity --no-ref-batch --no-torch-baseline --no-fast-mode --no-x3-mode --no-fwd --no-roofline"
for v in 1 2; do
  timeout 300 python bench.py $COMMON > $O/r06_bench_j_tmp.log 2>&1
  python - "$O/r06_bench_j_tmp.log" <<'PY'
import json, sys
l = [x for x in open(sys.argv[1]) if x.startswith('{')]
if not l:
    print("NO LINE"); print(open(sys.argv[1]).read()[-1500:])
else:
    d = json.loads(l[-1]); print(d['value'], d['ms_per_step'])
PY
done 2>&1 | tee $O/r06_bench_j.log
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_j -o j -- python $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-torch-baseline --no-ref-batch --no-x3-mode --no-fast-mode --no-parity --no-fwd --no-roofline > $O/r06_prof_j_run.log 2>&1
python $REPO/tools/prof_summary.py /tmp/prof_j 5 > $O/r06_train_ns6_b63_x3f_j.txt 2>&1
head -3 $O/r06_train_ns6_b63_x3f_j.txt | cut -c1-160
