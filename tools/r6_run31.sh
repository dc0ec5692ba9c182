mkdir -p gpurun_out/r6ac
python -m pytest tests -q -m gpu > gpurun_out/r6ac/pytest_full.log 2>&1; echo "full rc=$?"; tail -4 gpurun_out/r6ac/pytest_full.log
bash tools/collect_r06.sh > gpurun_out/r6ac/collect.log 2>&1; echo "collect rc=$?"; tail -3 gpurun_out/r6ac/collect.log
bash tools/headline_repeats.sh > gpurun_out/r06/r06_headline_repeats.log 2>&1; tail -6 gpurun_out/r06/r06_headline_repeats.log
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r6ac/bench_driver_args.json 2> gpurun_out/r6ac/bench_driver_args.err; echo "bench rc=$?"; cut -c1-600 gpurun_out/r6ac/bench_driver_args.json
