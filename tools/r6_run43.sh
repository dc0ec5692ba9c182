mkdir -p gpurun_out/r6al
timeout 1100 python -m pytest tests -q -m gpu > gpurun_out/r6al/pytest_full.log 2>&1; echo "full rc=$?"; tail -5 gpurun_out/r6al/pytest_full.log | cut -c1-300
