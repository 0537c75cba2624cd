#!/bin/bash
# Runs on the GPU box: the round's evidence in one call -- bench lines and the rocprofv3 passes behind profiles/.
#   usage: tools/r02_collect.sh <tag>
TAG=${1:-r02}
export TMPDIR=/tmp
python3 bench.py > gpurun_out/${TAG}_bench_b4096.json 2> gpurun_out/${TAG}_bench_b4096.err
echo "[collect] bench 4096 done"
python3 bench.py --in-flight 2 --no-cpu-baseline > gpurun_out/${TAG}_bench_b4096_f2.json 2> gpurun_out/${TAG}_bench_b4096_f2.err
python3 bench.py --in-flight 3 --no-cpu-baseline > gpurun_out/${TAG}_bench_b4096_f3.json 2> gpurun_out/${TAG}_bench_b4096_f3.err
echo "[collect] bench 4096 in-flight 2/3 done"
python3 bench.py --batch 65536 --steps 40 --warmup 5 --no-cpu-baseline > gpurun_out/${TAG}_bench_b65536.json 2> gpurun_out/${TAG}_bench_b65536.err
echo "[collect] bench 65536 done"
python3 bench.py --batch 65536 --gt 1 --steps 30 --warmup 5 > gpurun_out/${TAG}_bench_gt_sc1_b65536.json 2> gpurun_out/${TAG}_bench_gt1.err
python3 bench.py --batch 65536 --gt 3 --steps 30 --warmup 5 > gpurun_out/${TAG}_bench_gt_sc3_b65536.json 2> gpurun_out/${TAG}_bench_gt3.err
echo "[collect] bench gt done"
IGT_PMC_DTYPE=f64 IGT_PMC_BATCH=4096 tools/pmc_collect.sh ${TAG}_f64_b4096 > gpurun_out/${TAG}_collect_f64_b4096.log 2>&1
echo "[collect] pmc f64 4096 done"
IGT_PMC_DTYPE=f64 IGT_PMC_BATCH=65536 tools/pmc_collect.sh ${TAG}_f64_b65536 --batch 65536 --steps 8 --warmup 2 --no-cpu-baseline --no-secondary > gpurun_out/${TAG}_collect_f64_b65536.log 2>&1
echo "[collect] pmc f64 65536 done"
IGT_PMC_DTYPE=f32 IGT_PMC_BATCH=4096 tools/pmc_collect.sh ${TAG}_f32_b4096 --dtype f32 --steps 20 --warmup 3 --no-cpu-baseline --no-secondary > gpurun_out/${TAG}_collect_f32_b4096.log 2>&1
echo "[collect] pmc f32 4096 done"
IGT_PMC_DTYPE=f32 IGT_PMC_BATCH=65536 tools/pmc_collect.sh ${TAG}_f32_gt1_b65536 --dtype f32 --gt 1 --batch 65536 --steps 8 --warmup 2 --no-secondary > gpurun_out/${TAG}_collect_f32_gt1.log 2>&1
echo "[collect] pmc f32 gt 65536 done"
