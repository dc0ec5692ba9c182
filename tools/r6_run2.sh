mkdir -p gpurun_out/r6b
python -m pytest tests/test_gpu_fastscore.py tests/test_gpu_multirank.py tests/test_gpu_slot.py -x -q -m gpu > gpurun_out/r6b/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r6b/pytest.log
tail -15 gpurun_out/r6b/pytest.log
python bench.py > gpurun_out/r6b/bench.json 2> gpurun_out/r6b/bench.err; echo "bench rc=$?"
tail -3 gpurun_out/r6b/bench.err
head -c 2500 gpurun_out/r6b/bench.json
