mkdir -p gpurun_out/r04i
for f in 0 16384; do
  for c in lattice track; do
    IGT_DEV_FLAGS=$f python3 bench.py --batch 65536 --cand $c --steps 20 --warmup 3 --no-cpu-baseline --no-configs --no-secondary --in-flight 1 2>/dev/null | python3 tools/ab_line.py "flags $f $c B=65536 (hold $((f>>12)) -> default 8 when 0)"
  done
done
