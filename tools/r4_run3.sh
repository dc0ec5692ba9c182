#!/bin/bash
# development: the CLI + broker tests, broker throughput against tick size
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
python -m pytest tests/test_gpu_broker.py tests/test_gpu_parity.py tests/test_gpu_lazy.py tests/test_gpu_multirank.py -q -m gpu -s -k "broker or cli or multirank or two_ranks" > gpurun_out/r4_tests_b2.log 2>&1
grep -n "callers through\|broker:\|passed\|failed" gpurun_out/r4_tests_b2.log
for tf in 64 128 256 512; do
  JD_BROKER_TICK_FRAMES=$tf python -m pytest tests/test_gpu_broker.py -q -m gpu -s -k throughput 2>&1 | grep "callers through" | sed "s/^/tick $tf: /"
done | tee gpurun_out/r4_broker_ticks.log
