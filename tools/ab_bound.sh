export TMPDIR=/tmp; O=gpurun_out/r04b; mkdir -p $O
python -m pytest tests/test_gpu_parity.py -x -q -k "live_acceleration_rows or production_kernel_switches or tracking" > $O/tests.log 2>&1; echo "tests rc=$?"; tail -4 $O/tests.log
python tools/family_probe.py 4096 65536 2>&1 | grep -v amdgpu.ids | grep "f64" > $O/family_bound.txt
IGT_DEV_FLAGS=8388608 python tools/family_probe.py 4096 65536 2>&1 | grep -v amdgpu.ids | grep "f64.*track" > $O/family_nobound.txt
echo "--- with bound"; cat $O/family_bound.txt; echo "--- without"; cat $O/family_nobound.txt
