mkdir -p gpurun_out/r6aj
echo "== diag without JD_VERBOSE"; NOPLAIN=1 NB=3 timeout 70 python tools/rt_pipe_diag.py 8001 4 7 3 50 2>&1 | grep -v amdgpu.ids | tail -8 | cut -c1-300
echo "== pytest with JD_VERBOSE"; JD_VERBOSE=1 timeout 120 python -m pytest tests/test_gpu_random_topology.py -x -q -k "pipeline and 8001" > gpurun_out/r6aj/pytest.log 2>&1; echo "rc=$?"; grep -v "^$" gpurun_out/r6aj/pytest.log | grep -A40 "Captured stderr" | cut -c1-300 | head -60
