#!/bin/bash
# GPU box: gt-mode (value-net cost) bench lines for both network depths at two batch sizes
for sc in 1 3; do for b in 4096 65536; do
  python3 bench.py --gt $sc --batch $b --steps 10 --warmup 3 --no-cpu-baseline | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('gt sc=$sc B=$b', {k: round(v,4) for k,v in d['kernels_ms'].items()}, round(d['value']/1e6,3),'M/s')"
done; done
