#!/bin/bash
# GPU box: search-kernel time and VALU instruction count under the developer switches (IGT_DEV_FLAGS)
set -e
export TMPDIR=/tmp
for f in ${FLAGS:-0 1 2 3}; do
  export IGT_DEV_FLAGS=$f
  O=gpurun_out/dev$f; mkdir -p $O
  rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $O/pmc1 -- python3 bench.py --batch 65536 --steps 4 --warmup 2 --no-cpu-baseline > $O/bench.log 2>&1
  python3 tools/pmc_summarise.py $O | grep -E "search_fast\.(SQ|dur)" | sed "s/^/flags=$f /"
done
