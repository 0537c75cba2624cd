#!/bin/bash
# GPU box: the items per wave at the end of a queue whose index is fetched late (IGT_DEV_FLAGS bits 12-15; default 4): one solve
# at a time and four in flight, lattice and tracking
for h in 0 1 2 4 6 8 12 15; do
  f=$((h << 12))
  IGT_DEV_FLAGS=$f python3 bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-configs 2>/dev/null | python3 tools/ab_line.py "late items per wave $h"
done
