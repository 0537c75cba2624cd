export TMPDIR=/tmp; O=gpurun_out/r04c; mkdir -p $O
python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -q -x -k "other_shapes or live_acceleration or cartesian_row or oracle_order or advertised or search_and_emit or switches or big_batch" > $O/tests.log 2>&1; echo "tests rc=$?"; tail -5 $O/tests.log
