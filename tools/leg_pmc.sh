#!/bin/bash
# HBM traffic of k_search in one of bench.py's workloads: tools/leg_pmc.sh c2|hyps|c512|north|c3|clg  (GPU box, through gpurun).
# Two PMC passes, each with --kernel-trace only; summary -> gpurun_out/prof_<leg>/pmc_summary.json
set -u
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
export JD_DEV=1
export JD_BENCH_NO_LAZY=1      # the clg leg: without its search-driven part
LEG=${1:-north}
OUT=gpurun_out/prof_$LEG
rm -rf "$OUT"; mkdir -p "$OUT"
i=0
for set in "FETCH_SIZE SQ_WAVES" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"; do
    i=$((i + 1))
    timeout 900 rocprofv3 --kernel-trace --pmc $set --output-format csv -d "$OUT/pmc$i" -- python tools/run_leg.py $LEG ${LEG_PASSES:-2} < /dev/null > "$OUT/pmc$i.log" 2>&1
    grep '^{"workload"' "$OUT/pmc$i.log" | tail -1 > "$OUT/leg_under_pmc$i.json"
done
python tools/pmc_summary.py "$OUT" > "$OUT/pmc_summary.json"
python - "$OUT" "$LEG" <<'PY'
import json, os, sys
sys.path.insert(0, os.getcwd())
import bench
d = json.load(open(sys.argv[1] + "/pmc_summary.json"))
tot_f = tot_w = 0.0
for k, v in d.items():
    if "k_search" in k or "k_resident" in k or "k_slot" in k:
        print(k, {c: (x["launches"], round(x["mean"], 1), round(x["max"], 1)) for c, x in v.items()})
        tot_f += v["FETCH_SIZE"]["mean"] * v["FETCH_SIZE"]["launches"]
        tot_w += v["WRITE_SIZE"]["mean"] * v["WRITE_SIZE"]["launches"]
# tools/run_leg.py <leg> N = decode calls of the same batch (N, one more with two batches in flight): HBM bytes of ONE
# pass over the batch, all k_search launches, calibrated (bench.calibrated_traffic: the counters tally scattered accesses
# and writes exactly and wide coalesced reads at half - the launch's wide reads are its instance records, 80 B per instance
# processed).  With two batches in flight the calls have also searched part of the batch behind the last one - as much
# of a batch as a batch is ahead when its turn comes: that many batches' worth of search were counted.
leg = json.load(open(sys.argv[1] + "/leg_under_pmc1.json"))
# (the records the kernel really READ - its own counter; tot_insts_in is the reference's instance count, which includes
# candidates that never become a record: round 5 priced 11.7 k records per configs[1] frame where ~5 k are read)
rec_bytes = 80.0 if "1-6 emitting" not in leg.get("workload", "") else 144.0
wide = rec_bytes * leg["per_stream_frame"]["tot_recs_read"] * leg["frames_per_step"]
np_ = float(leg.get("decode_calls", 2)) + (leg.get("searched_ahead_frames", 0) / float(leg["frames_per_step"]) if leg.get("batches_in_flight", 1) == 2 else 0.0)
out = {"leg": sys.argv[2], "passes": np_, "k_search_hbm_bytes_per_pass": bench.calibrated_traffic(tot_f / np_, tot_w / np_, wide),
       "uncalibrated_2xFETCH_plus_WRITE_bytes_per_pass": (2.0 * tot_f + tot_w) * 1024.0 / np_,
       "FETCH_SIZE_KiB_per_pass": tot_f / np_, "WRITE_SIZE_KiB_per_pass": tot_w / np_, "wide_read_bytes_per_pass": wide, "wide_reads_are": "record bytes x tot_recs_read (records the kernel read)",
       "algorithmic_bytes_per_pass": leg["roofline"]["algorithmic_bytes_per_launch"] * leg["roofline"]["launches_per_step"],
       "frames_per_pass": leg["frames_per_step"],
       "search_ms_under_pmc": leg["search_ms"], "source_hash": bench.kernel_source_hash(),
       "source": "tools/leg_pmc.sh: rocprofv3 --kernel-trace --pmc, FETCH_SIZE and WRITE_SIZE in separate runs of `python tools/run_leg.py %s 2`" % sys.argv[2]}
json.dump(out, open(sys.argv[1] + "/leg_traffic.json", "w"), indent=1)
print(out)
PY
