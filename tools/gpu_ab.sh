#!/bin/bash
# Runs on the GPU box: a subset of the GPU tests, bench.py with another build of the library beside the tree's (tools/ab_lib.sh),
# and a kernel trace of the tree's build with one solve in flight (per-kernel durations of the prelude).
#   usage: tools/gpu_ab.sh <tag> <other.so> "<pytest -k expression>"
export TMPDIR=/tmp; O=gpurun_out/${1:-ab}; mkdir -p $O
timeout -k 10 600 python -m pytest tests -q -x -m gpu -k "$3" > $O/tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -4 $O/tests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 400 bash tools/ab_lib.sh $2 > $O/ab.txt 2>&1; cat $O/ab.txt
cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof -o run -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-secondary --no-configs --in-flight 1 --settle-ms 0 --steps 20 --warmup 3 > $GRAFT_REPO_ROOT/$O/prof.log 2>&1
cd $GRAFT_REPO_ROOT && find $O/prof -name "*kernel_stats.csv" | head -1 | xargs -r head -6 | cut -c1-60,300-420
