mkdir -p gpurun_out/r6af
python -m pytest tests/test_gpu_refgolden.py tests/test_gpu_random_topology.py tests/test_refgolden_cpu.py -q > gpurun_out/r6af/pytest.log 2>&1; echo "rc=$?"; tail -4 gpurun_out/r6af/pytest.log
