#!/bin/bash
# GPU box: per-kernel durations of the gt-mode solve (rocprofv3 kernel trace), both network depths, B = 65536
export TMPDIR=/tmp
for sc in 1 3; do
  O=gpurun_out/gt_trace_sc$sc; rm -rf $O; mkdir -p $O
  rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python3 bench.py --gt $sc --batch ${BATCH:-65536} --steps 10 --warmup 3 --no-cpu-baseline > $O/log.txt 2>&1
  python3 - <<PY
import csv,glob
for f in glob.glob("$O/**/*kernel_stats.csv",recursive=True):
    for r in list(csv.DictReader(open(f)))[:6]: print("sc$sc", r["Name"][:48], r["Calls"], round(float(r["AverageNs"])/1e3,1), "us")
PY
done
