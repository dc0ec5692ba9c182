mkdir -p gpurun_out/r6w
python -m pytest tests/test_gpu_parity.py -x -q -k "own_state_numbering" > gpurun_out/r6w/pytest_new.log 2>&1; echo "new rc=$?"; tail -5 gpurun_out/r6w/pytest_new.log
python -m pytest tests -x -q -m gpu > gpurun_out/r6w/pytest_full.log 2>&1; echo "full rc=$?"; tail -3 gpurun_out/r6w/pytest_full.log
bash tools/collect_r06.sh > gpurun_out/r6w/collect.log 2>&1; echo "collect rc=$?"; tail -3 gpurun_out/r6w/collect.log
bash tools/headline_repeats.sh > gpurun_out/r06/r06_headline_repeats.log 2>&1; tail -6 gpurun_out/r06/r06_headline_repeats.log
