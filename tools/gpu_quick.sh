export TMPDIR=/tmp; O=gpurun_out/${1:-r04h}; mkdir -p $O
python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_fuzz.py -q -x -k "switches or search_and_emit or tracking or f32" > $O/tests.log 2>&1; echo "tests rc=$?"; tail -6 $O/tests.log
python tools/family_probe.py 4096 65536 2>&1 | grep "track" > $O/family.txt; cat $O/family.txt
IGT_DEV_FLAGS=8388608 python tools/family_probe.py 4096 65536 2>&1 | grep "f32.*track" > $O/family_nobound.txt; cat $O/family_nobound.txt
