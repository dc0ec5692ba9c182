mkdir -p gpurun_out/r6ai
timeout 150 python -m pytest tests/test_gpu_random_topology.py -x -q -k "pipeline and 8001" > gpurun_out/r6ai/pytest.log 2>&1; echo "rc=$?"; grep -v "^$" gpurun_out/r6ai/pytest.log | tail -40 | cut -c1-400
