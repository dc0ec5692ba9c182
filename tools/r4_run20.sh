#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
for fc in 64 96 128; do echo "== 16 threads, free CUs $fc"; JD_RES_FREE_CUS=$fc timeout 600 python tools/broker_cli_bench.py 16 2>&1 | grep resident; done
for fc in 51 64 96; do echo "== 32 threads, free CUs $fc"; JD_RES_FREE_CUS=$fc timeout 600 python tools/broker_cli_bench.py 32 2>&1 | grep resident; done
for fc in 40 128; do echo "== 4 threads, free CUs $fc"; JD_RES_FREE_CUS=$fc timeout 600 python tools/broker_cli_bench.py 4 2>&1 | grep resident; done
