#!/bin/bash
# GPU box: the driver's 20-step command with and without the drain hint of bench.py (IGT_BENCH_NO_DRAIN_HINT=1), alternating
for i in 1 2 3 4; do
  IGT_BENCH_NO_DRAIN_HINT=1 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-configs --no-secondary 2>/dev/null | python3 tools/ab_line.py "no hint  run $i"
  python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-configs --no-secondary 2>/dev/null | python3 tools/ab_line.py "hint     run $i"
done
