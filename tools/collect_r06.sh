#!/bin/bash
# Round-6 profile evidence, collected on the GPU box in one go (through gpurun, from the repo root):
#   1. rocprofv3 --kernel-trace --stats of the bench command itself (the slot pipeline: ONE k_slot launch spans the timed region)
#   2. SQ / instruction-mix / L1 / L2 PMC passes of the slot kernel as a plain launch (k_slot_batch over configs[2]'s 512 utterances:
#      under --pmc kernels run one after the other, and the pipeline's k_slot waits for the kernels beside it)
#   3. tools/leg_pmc.sh <leg>: FETCH_SIZE / WRITE_SIZE passes of every workload -> calibrated traffic (+ the hash of the kernel sources)
#   4. python bench.py: the full line with its legs, reading the traffic files just written
# Results land in gpurun_out/r06/ with the names they get in profiles/.
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
R=gpurun_out/r06
rm -rf "$R"; mkdir -p "$R"
OUT=gpurun_out/prof
rm -rf "$OUT"; mkdir -p "$OUT"
BENCH="python bench.py --no-cpu-baseline --no-extra-legs"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- $BENCH < /dev/null > "$OUT/stats.log" 2>&1
f=$(find "$OUT/stats" -name '*kernel_stats.csv' | head -1)
[ -n "$f" ] && cp "$f" "$R/r06_c2_kernel_stats.csv"
grep "^{\"metric\"" "$OUT/stats.log" | tail -1 > "$R/r06_c2_bench_under_rocprof.json"
i=0
for set in "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_ATOMIC_WITH_RET_REQ_sum" \
           "TCP_TCC_ATOMIC_WITHOUT_RET_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_GATE_EN1_sum" \
           "TCC_REQ_sum TCC_ATOMIC_sum TCC_TAG_STALL_sum" \
           "SQ_WAIT_INST_LDS SQ_INSTS_FLAT SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
    i=$((i + 1))
    timeout 600 rocprofv3 --kernel-trace --pmc $set --output-format csv -d "$OUT/pmc$i" -- python tools/run_leg.py c512slot 2 < /dev/null > "$OUT/pmc$i.log" 2>&1
done
python tools/pmc_summary.py "$OUT" > "$R/r06_slot_pmc_summary.json"
for l in ${LEGS:-c512slot c2 hyps north c3 clg}; do
    p=2; case $l in c2|hyps) p=6;; esac                               # (two batches in flight: more calls, smaller edge)
    LEG_PASSES=$p tools/leg_pmc.sh $l > "$R/leg_$l.log" 2>&1
    cp gpurun_out/prof_$l/leg_traffic.json "$R/r06_${l}_traffic.json"
    cp gpurun_out/prof_$l/leg_traffic.json "profiles/r06_${l}_traffic.json"     # (on the box: bench.py reads them below)
    cp gpurun_out/prof_$l/pmc_summary.json "$R/r06_${l}_leg_pmc_summary.json"
done
python bench.py > "$R/r06_bench_full_with_legs.json" 2> "$R/bench.err"
tail -c 400 "$R/r06_bench_full_with_legs.json"
ls -la "$R"
