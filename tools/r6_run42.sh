mkdir -p gpurun_out/r6ak
timeout 200 python -m pytest tests/test_gpu_random_topology.py -x -q -k "pipeline" > gpurun_out/r6ak/pytest.log 2>&1; echo "rc=$?"; grep -v "^$" gpurun_out/r6ak/pytest.log | tail -25 | cut -c1-300
