mkdir -p gpurun_out/r6ae
python -m pytest tests/test_gpu_random_topology.py -q -s > gpurun_out/r6ae/pytest_rt.log 2>&1; echo "rt rc=$?"; grep "random topologies\|passed\|failed\|Error\|assert" gpurun_out/r6ae/pytest_rt.log | head -20
