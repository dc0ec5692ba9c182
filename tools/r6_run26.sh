mkdir -p gpurun_out/r6x
python -m pytest tests/test_gpu_parity.py -x -q -k "own_state_numbering" > gpurun_out/r6x/pytest_new.log 2>&1; echo "new rc=$?"; tail -5 gpurun_out/r6x/pytest_new.log
python -m pytest tests -q -m gpu > gpurun_out/r6x/pytest_full.log 2>&1; echo "full rc=$?"; tail -5 gpurun_out/r6x/pytest_full.log
