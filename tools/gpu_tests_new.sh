export TMPDIR=/tmp; O=gpurun_out/r04c; mkdir -p $O
python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -q -k "other_shapes or live_acceleration or cartesian_row or oracle_order or advertised" > $O/tests.log 2>&1; echo "tests rc=$?"; tail -30 $O/tests.log
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_driver.json 2> $O/bench_driver.err; echo "bench rc=$?"
