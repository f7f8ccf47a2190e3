#!/bin/bash
# GPU session A of round 2: full -m gpu suite (incl. the new full-size parity tests), the default bench line, an x3 bench line and a
# rocprofv3 kernel summary of the bench command.  Everything lands in gpurun_out/.
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
rm -f gpurun_out/parity_report.jsonl
nproc > gpurun_out/r02_host.txt; lscpu | grep -E "Model name|^CPU\(s\)|Thread|Socket" >> gpurun_out/r02_host.txt; free -g | head -2 >> gpurun_out/r02_host.txt
timeout 1500 python -m pytest tests -m gpu -q -rf --durations=15 > gpurun_out/r02_pytest_a.log 2>&1
tail -5 gpurun_out/r02_pytest_a.log
timeout 600 python bench.py --steps 8 --warmup 2 > gpurun_out/r02_bench_a.log 2>&1
tail -1 gpurun_out/r02_bench_a.log
timeout 600 python bench.py --prec x3 --batch 24 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r02_bench_a_x3.log 2>&1
tail -1 gpurun_out/r02_bench_a_x3.log
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_a -o train -- python "$GRAFT_REPO_ROOT/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --no-roofline > "$GRAFT_REPO_ROOT/gpurun_out/r02_prof_a.log" 2>&1
python "$GRAFT_REPO_ROOT/tools/prof_summary.py" /tmp/prof_a 3 > "$GRAFT_REPO_ROOT/gpurun_out/r02_prof_a.txt" 2>&1
head -30 "$GRAFT_REPO_ROOT/gpurun_out/r02_prof_a.txt"
