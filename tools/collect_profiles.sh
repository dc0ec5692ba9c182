#!/bin/bash
# Collects the round's profile evidence on the GPU box (run through gpurun from the repo root):
#   1. rocprofv3 --kernel-trace --stats of the bench command itself
#   2. PMC passes, each in its own run with --kernel-trace only (never with other trace domains)
#   3. tools/pmc_summary.py -> gpurun_out/prof/pmc_summary.json
# Copy the results you want judged from gpurun_out/prof/ into profiles/.
set -u
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
OUT=gpurun_out/prof
rm -rf "$OUT"; mkdir -p "$OUT"
BENCH="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra-legs"
ONE="python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extra-legs"   # (the warm-up batch takes the decoder's one-off 32-frame probe launch)

timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- $BENCH < /dev/null > "$OUT/stats.log" 2>&1
f=$(find "$OUT/stats" -name '*kernel_stats.csv' | head -1)
[ -n "$f" ] && cp "$f" "$OUT/kernel_stats.csv"
grep "^{\"metric\"" "$OUT/stats.log" | tail -1 > "$OUT/bench_under_rocprof.json"

i=0
for set in "FETCH_SIZE SQ_WAVES" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE"; do
    i=$((i + 1))
    timeout 600 rocprofv3 --kernel-trace --pmc $set --output-format csv -d "$OUT/pmc$i" -- $ONE < /dev/null > "$OUT/pmc$i.log" 2>&1
done
python tools/pmc_summary.py "$OUT" > "$OUT/pmc_summary.json"
ls -la "$OUT"
