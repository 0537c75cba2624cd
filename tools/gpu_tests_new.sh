export TMPDIR=/tmp; O=gpurun_out/r04d; mkdir -p $O
python -m pytest tests/test_gpu_parity.py -q -x -k "emit_in_pieces or search_and_emit or switches" > $O/tests.log 2>&1; echo "tests rc=$?"; tail -15 $O/tests.log
python tools/family_probe.py 4096 65536 2>&1 | grep "f64" > $O/family.txt; cat $O/family.txt
