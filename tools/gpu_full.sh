export TMPDIR=/tmp; O=gpurun_out/${1:-r04e}; mkdir -p $O
python -m pytest tests -q -x -m gpu > $O/tests.log 2>&1; echo "tests rc=$?"; tail -6 $O/tests.log
python bench.py --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err; echo "bench rc=$?"
