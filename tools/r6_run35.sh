mkdir -p gpurun_out/r6ag
python -m pytest tests/test_gpu_random_topology.py -q -s -k larger > gpurun_out/r6ag/pytest.log 2>&1; echo "rc=$?"; grep "larger random\|passed\|failed\|Error\|assert" gpurun_out/r6ag/pytest.log | head -20
