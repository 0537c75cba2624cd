export TMPDIR=/tmp; O=gpurun_out/${1:-r04f}; mkdir -p $O
python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -q -x -k "live_acceleration or switches or config2 or solve_matches or tracking_candidates" > $O/tests.log 2>&1; echo "tests rc=$?"; tail -4 $O/tests.log
bash tools/ab_lib.sh tools/libigtmpc_r04a.bin --steps 40 --warmup 5 --no-cpu-baseline --no-configs 2>&1 | tee $O/ab.txt
