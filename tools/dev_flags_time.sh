#!/bin/bash
# GPU box: kernel times of bench.py under developer switches (IGT_DEV_FLAGS, IGT_DEV_SLOTS, IGT_DEV_STATIC)
for f in ${FLAGS:-0}; do
 for sl in ${SLOTS:-0}; do
  for sh in ${SHARES:-0.7}; do
   for b in ${BATCHES:-4096 65536}; do
    IGT_DEV_FLAGS=$f IGT_DEV_SLOTS=$sl IGT_DEV_STATIC=$sh python3 bench.py --batch $b --steps 20 --warmup 5 --no-cpu-baseline | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('flags=$f slots=$sl static=$sh B=$b', {k: round(v,4) for k,v in d['kernels_ms'].items()}, round(d['value']/1e6,3),'M/s')"
   done
  done
 done
done
