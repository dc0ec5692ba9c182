#!/bin/bash
# Round-4 profile evidence, collected on the GPU box in one go (through gpurun, from the repo root):
#   1. rocprofv3 --kernel-trace --stats of the bench command itself (two batches in flight: three warm-up steps fill the pipeline)
#   2. the SQ / instruction-mix PMC passes of the same command (each in its own run, --kernel-trace only)
#   3. tools/leg_pmc.sh <leg>: FETCH_SIZE / WRITE_SIZE passes of every workload -> calibrated traffic (+ the hash of the kernel sources)
#   4. python bench.py: the full line with its legs, reading the traffic files just written
# Results land in gpurun_out/r04/ with the names they get in profiles/.
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
R=gpurun_out/r04
rm -rf "$R"; mkdir -p "$R"
OUT=gpurun_out/prof
rm -rf "$OUT"; mkdir -p "$OUT"
BENCH="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra-legs"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- $BENCH < /dev/null > "$OUT/stats.log" 2>&1
f=$(find "$OUT/stats" -name '*kernel_stats.csv' | head -1)
[ -n "$f" ] && cp "$f" "$R/r04_c2_kernel_stats.csv"
grep "^{\"metric\"" "$OUT/stats.log" | tail -1 > "$R/r04_c2_bench_under_rocprof.json"
i=0
for set in "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE" \
           "FETCH_SIZE SQ_WAVES" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"; do
    i=$((i + 1))
    # (counters: rocprofv3 runs kernels one after the other under --pmc, and the resident kernel of the default run waits for the scoring
    # kernels beside it - the passes see the same batches with two of them in flight, one k_search launch per step)
    timeout 600 rocprofv3 --kernel-trace --pmc $set --output-format csv -d "$OUT/pmc$i" -- $BENCH --pipeline-depth 0 < /dev/null > "$OUT/pmc$i.log" 2>&1
done
python tools/pmc_summary.py "$OUT" > "$R/r04_c2_pmc_summary.json"
for l in ${LEGS:-c2 hyps c512 north c3 clg}; do
    p=2; case $l in c2|hyps) p=6;; esac                               # (two batches in flight: more calls, smaller edge)
    LEG_PASSES=$p tools/leg_pmc.sh $l > "$R/leg_$l.log" 2>&1
    cp gpurun_out/prof_$l/leg_traffic.json "$R/r04_${l}_traffic.json"
    cp gpurun_out/prof_$l/leg_traffic.json "profiles/r04_${l}_traffic.json"     # (on the box: bench.py reads them below)
    cp gpurun_out/prof_$l/pmc_summary.json "$R/r04_${l}_leg_pmc_summary.json"
done
python bench.py > "$R/r04_bench_full_with_legs.json" 2> "$R/bench.err"
tail -c 400 "$R/r04_bench_full_with_legs.json"
ls -la "$R"
