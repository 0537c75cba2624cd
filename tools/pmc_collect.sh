#!/bin/bash
# Runs on the GPU box (through gpurun): kernel-trace stats + the PMC passes of MI355X_MICROARCH.md's HBM section,
# each counter group in its own pass, then summarises them into gpurun_out/<tag>_pmc.json.
#   usage: tools/pmc_collect.sh <tag> [bench.py args...]
set -e
TAG=${1:-pmc}; shift || true
ARGS="${@:---steps 20 --warmup 3 --no-cpu-baseline --no-secondary --no-configs --in-flight 1 --settle-ms 0}"
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 bench.py $ARGS > $OUT/bench_stats.log 2>&1
echo "[pmc_collect] stats pass done"
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" \
           "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "GRBM_GUI_ACTIVE SQ_WAIT_ANY"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/pmc$i -- python3 bench.py $ARGS > $OUT/bench_pmc$i.log 2>&1
  echo "[pmc_collect] pmc pass $i ($grp) done"
done
python3 tools/pmc_summarise.py $OUT > $OUT/summary.json
cat $OUT/summary.json
