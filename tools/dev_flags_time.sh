#!/bin/bash
# GPU box: kernel times of bench.py under developer switches (IGT_DEV_FLAGS; see igt_device.h) at several batch sizes
#   FLAGS="0 16" BATCHES="1024 4096" tools/dev_flags_time.sh
for f in ${FLAGS:-0}; do
  for b in ${BATCHES:-4096 65536}; do
    IGT_DEV_FLAGS=$f python3 bench.py --batch $b --steps 20 --warmup 5 --no-cpu-baseline | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('flags=$f B=$b', {k: round(v,4) for k,v in d['kernels_ms'].items()}, round(d['value']/1e6,3),'M/s')"
  done
done
