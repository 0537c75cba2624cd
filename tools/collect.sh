#!/bin/bash
# Runs on the GPU box: the round's evidence in one call -- the rocprofv3 passes behind profiles/, then bench lines and probes.
#   usage: tools/collect.sh <tag> [round=r04] [what=all|pmc|bench|probes]     then, back home:  python tools/publish.py <tag> <round>
TAG=${1:-r04}
ROUND=${2:-r04}
WHAT=${3:-all}
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
COMMON="--no-cpu-baseline --no-secondary --no-configs --in-flight 1 --settle-ms 0"
if [ "$WHAT" = all ] || [ "$WHAT" = pmc ]; then
  # 0. issue rates of the instructions the rollout is made of (quoted by the counter summaries)
  hipcc --offload-arch=gfx950 -O3 tools/valu_microbench.hip -o /tmp/valu_mb > $O/${TAG}_valu_mb_build.log 2>&1 && /tmp/valu_mb > $O/${TAG}_valu_microbench.txt 2>&1
  echo "[collect] valu microbench done"
  # 1. counter passes first, published into this box's profiles/ so that the bench lines below can quote them
  #    (bench.py only quotes a profile whose source_hash equals the hash of the kernel sources it runs on)
  export IGT_PMC_MICROBENCH=$O/${TAG}_valu_microbench.txt
  IGT_PMC_DTYPE=f64 IGT_PMC_BATCH=4096 tools/pmc_collect.sh ${TAG}_f64_b4096 --steps 20 --warmup 3 $COMMON > $O/${TAG}_collect_f64_b4096.log 2>&1
  echo "[collect] pmc f64 4096 done"
  IGT_PMC_DTYPE=f64 IGT_PMC_BATCH=4096 tools/pmc_collect.sh ${TAG}_f64_track_b4096 --cand track --steps 20 --warmup 3 $COMMON > $O/${TAG}_collect_f64_track_b4096.log 2>&1
  echo "[collect] pmc f64 tracking 4096 done"
  IGT_PMC_DTYPE=f64 IGT_PMC_BATCH=65536 tools/pmc_collect.sh ${TAG}_f64_b65536 --batch 65536 --steps 8 --warmup 2 $COMMON > $O/${TAG}_collect_f64_b65536.log 2>&1
  echo "[collect] pmc f64 65536 done"
  IGT_PMC_DTYPE=f32 IGT_PMC_BATCH=4096 tools/pmc_collect.sh ${TAG}_f32_b4096 --dtype f32 --steps 20 --warmup 3 $COMMON > $O/${TAG}_collect_f32_b4096.log 2>&1
  echo "[collect] pmc f32 4096 done"
  IGT_PMC_DTYPE=f32 IGT_PMC_BATCH=65536 tools/pmc_collect.sh ${TAG}_f32_gt1_b65536 --dtype f32 --gt 1 --batch 65536 --steps 8 --warmup 2 $COMMON > $O/${TAG}_collect_f32_gt1.log 2>&1
  IGT_PMC_DTYPE=f64 IGT_PMC_BATCH=65536 tools/pmc_collect.sh ${TAG}_f64_gt1_b65536 --dtype f64 --gt 1 --batch 65536 --steps 6 --warmup 2 $COMMON > $O/${TAG}_collect_f64_gt1.log 2>&1
  echo "[collect] pmc gt 65536 done"
  python3 tools/publish.py ${TAG} ${ROUND} > $O/${TAG}_publish_on_box.log 2>&1
  echo "[collect] counter summaries published on the box"
fi
if [ "$WHAT" = all ] || [ "$WHAT" = bench ]; then
  # 2. bench lines: the driver's command (20 steps, 5 warm-up) and the long one, then the other configurations on their own
  python3 bench.py --steps 20 --warmup 5 > $O/${TAG}_bench_b4096_driver_args.json 2> $O/${TAG}_bench_b4096_driver_args.err
  python3 bench.py > $O/${TAG}_bench_b4096.json 2> $O/${TAG}_bench_b4096.err
  echo "[collect] bench 4096 done"
  python3 bench.py --batch 65536 --steps 40 --warmup 5 --no-cpu-baseline --in-flight 1 > $O/${TAG}_bench_b65536.json 2> $O/${TAG}_bench_b65536.err
  python3 bench.py --gt 1 --steps 30 --warmup 5 --in-flight 1 > $O/${TAG}_bench_gt_sc1_b65536.json 2> $O/${TAG}_bench_gt1.err
  python3 bench.py --gt 3 --steps 30 --warmup 5 --in-flight 1 > $O/${TAG}_bench_gt_sc3_b65536.json 2> $O/${TAG}_bench_gt3.err
  echo "[collect] bench 65536 / gt done"
  for F in 1 2 3 4; do
    python3 bench.py --in-flight $F --no-cpu-baseline --no-configs 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('in-flight $F: f64 %.3f ms/step (%.2f M solves/s)   f32 %.3f ms/step (%.2f M solves/s)' % (d['ms_per_step'], d['value']/1e6, d['f32_path']['ms_per_step'], d['f32_path']['value']/1e6))"
  done > $O/${TAG}_inflight_sweep.txt
  echo "[collect] in-flight sweep done"
fi
if [ "$WHAT" = all ] || [ "$WHAT" = probes ]; then
  hipcc --offload-arch=gfx950 -O3 tools/mfma64_microbench.hip -o /tmp/mfma64_mb > $O/${TAG}_mfma64_build.log 2>&1 && /tmp/mfma64_mb > $O/${TAG}_mfma64_microbench.txt 2>&1
  python3 tools/f64_probe.py 4096 32768 2>&1 | grep -v amdgpu.ids > $O/${TAG}_f64_probe.txt
  python3 tools/f64_probe.py --flags=0,8 65536 2>&1 | grep -v amdgpu.ids >> $O/${TAG}_f64_probe.txt
  python3 tools/family_probe.py 2>&1 | grep -v amdgpu.ids > $O/${TAG}_family_probe.txt
  python3 tools/latency_probe.py 2>&1 | grep -v amdgpu.ids > $O/${TAG}_latency.txt
  python3 tools/f32_margin_probe.py 2048 2>&1 | grep -v amdgpu.ids > $O/${TAG}_f32_margin.txt
  python3 tools/closed_loop_probe.py f64 > $O/${TAG}_closed_loop.txt 2>&1
  python3 tools/closed_loop_probe.py f64 40 > $O/${TAG}_closed_loop_n40.txt 2>&1
  python3 tools/closed_loop_scale.py 2>&1 | grep -v amdgpu.ids > $O/${TAG}_closed_loop_scale.txt
  python3 tools/envelope_sweep.py 2>&1 | grep '^N=' > $O/${TAG}_envelope_sweep.txt
  { python3 tools/closed_loop_breakdown.py track; python3 tools/closed_loop_breakdown.py ramp_hold; python3 tools/closed_loop_breakdown.py track 40; python3 tools/closed_loop_breakdown.py track 20 1024; python3 tools/closed_loop_breakdown.py track 20 256 2; } 2>&1 | grep -v "^sc \|amdgpu.ids" > $O/${TAG}_closed_loop_breakdown.txt
  echo "[collect] probes done"
fi
