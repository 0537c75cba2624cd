#!/bin/bash
# Runs on the GPU box: a subset of the GPU tests, then bench.py with another build of the library beside the tree's (tools/ab_lib.sh).
#   usage: tools/gpu_ab.sh <tag> <other.so> "<pytest -k expression>"
export TMPDIR=/tmp; O=gpurun_out/${1:-ab}; mkdir -p $O
timeout -k 10 600 python -m pytest tests -q -x -m gpu -k "$3" > $O/tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -4 $O/tests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 400 bash tools/ab_lib.sh $2 > $O/ab.txt 2>&1; cat $O/ab.txt
