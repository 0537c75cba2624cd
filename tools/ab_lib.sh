#!/bin/bash
# A/B on one box: bench.py with another build of the library (IGT_LIB_PATH) against the tree's, alternating.
#   usage: tools/ab_lib.sh <other.so> [bench args]
OTHER=$1; shift
ARGS="${@:---steps 40 --warmup 5 --no-cpu-baseline --no-configs --no-secondary}"
cat > /tmp/ab_line.py <<'PY'
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%s: %.4f ms/step  %.2f M solves/s   search %.4f emit %.4f   one at a time %.2f M   tracking %.2f M (search %.4f)' % (
    sys.argv[1], d['ms_per_step'], d['value'] / 1e6, d['kernels_ms']['search'], d['kernels_ms']['emit'],
    d.get('one_solve_in_flight', {}).get('value', 0) / 1e6, d.get('tracking_family', {}).get('value', 0) / 1e6,
    d.get('tracking_family', {}).get('kernels_ms', {}).get('search', 0)))
PY
for i in 1 2 3; do
  for which in other tree; do
    if [ $which = other ]; then export IGT_LIB_PATH=$OTHER; else unset IGT_LIB_PATH; fi
    python3 bench.py $ARGS 2>/dev/null | python3 /tmp/ab_line.py "$which run $i"
  done
done
