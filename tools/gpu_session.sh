#!/bin/bash
# round 6, session 15: kernel trace with the precast form
cd "$GRAFT_REPO_ROOT" || exit 1
REPO="$GRAFT_REPO_ROOT"; O=$REPO/gpurun_out; mkdir -p $O
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_o -o o -- python $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-torch-baseline --no-ref-batch --no-x3-mode --no-fast-mode --no-parity --no-fwd --no-roofline > $O/r06_prof_o_run.log 2>&1
python $REPO/tools/prof_summary.py /tmp/prof_o 5 > $O/r06_train_ns6_b63_x3f_o.txt 2>&1
head -3 $O/r06_train_ns6_b63_x3f_o.txt | cut -c1-150; grep "ln_bwd\|rowscale_cast\|ln_dgb\|ln_fwd" $O/r06_train_ns6_b63_x3f_o.txt | cut -c1-150
