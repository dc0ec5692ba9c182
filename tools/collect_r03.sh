#!/bin/bash
# Round-3 profile evidence, collected on the GPU box in one go (through gpurun, from the repo root):
#   tools/collect_profiles.sh   rocprofv3 --kernel-trace --stats of the bench command + the SQ / instruction-mix PMC passes (configs[1])
#   tools/leg_pmc.sh <leg>      FETCH_SIZE / WRITE_SIZE passes of every workload -> calibrated traffic (+ the hash of the kernel sources)
#   python bench.py             the full line with its legs, reading the traffic files just written
# Results land in gpurun_out/r03/ with the names they get in profiles/.
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
R=gpurun_out/r03
rm -rf "$R"; mkdir -p "$R"
tools/collect_profiles.sh > "$R/collect.log" 2>&1
cp gpurun_out/prof/kernel_stats.csv "$R/r03_c2_kernel_stats.csv"
cp gpurun_out/prof/bench_under_rocprof.json "$R/r03_c2_bench_under_rocprof.json"
cp gpurun_out/prof/pmc_summary.json "$R/r03_c2_pmc_summary.json"
for l in c2 north c3 clg; do
    tools/leg_pmc.sh $l > "$R/leg_$l.log" 2>&1
    cp gpurun_out/prof_$l/leg_traffic.json "$R/r03_${l}_traffic.json"
    cp gpurun_out/prof_$l/leg_traffic.json "profiles/r03_${l}_traffic.json"     # (on the box: bench.py reads them below)
    cp gpurun_out/prof_$l/pmc_summary.json "$R/r03_${l}_leg_pmc_summary.json"
done
python bench.py > "$R/r03_bench_full_with_legs.json" 2> "$R/bench.err"
tail -c 600 "$R/r03_bench_full_with_legs.json"
