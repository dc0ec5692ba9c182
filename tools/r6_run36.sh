mkdir -p gpurun_out/r6ah
python -m pytest tests/test_gpu_random_topology.py -q -s -k pipeline > gpurun_out/r6ah/pytest.log 2>&1; echo "rc=$?"; grep "passed\|failed\|Error\|assert\|skipp" gpurun_out/r6ah/pytest.log | head -20
