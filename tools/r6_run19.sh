mkdir -p gpurun_out/r6t
python -m pytest tests -x -q -m gpu > gpurun_out/r6t/pytest_full.log 2>&1; echo "full rc=$?"; tail -3 gpurun_out/r6t/pytest_full.log
bash tools/collect_r06.sh > gpurun_out/r6t/collect.log 2>&1; echo "collect rc=$?"; tail -3 gpurun_out/r6t/collect.log
bash tools/headline_repeats.sh > gpurun_out/r06/r06_headline_repeats.log 2>&1; tail -6 gpurun_out/r06/r06_headline_repeats.log
