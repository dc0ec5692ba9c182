#!/bin/bash
# exploration: what saturates when every CU runs the search (the 512-utterance leg)?  a wider set of counters, one pass each
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
export JD_DEV=1
mkdir -p gpurun_out
OUT=gpurun_out/prof_explore
rm -rf "$OUT"; mkdir -p "$OUT"
rocprofv3 --list-avail 2>/dev/null | grep -oE "\b(TA|TD|TCP|TCC|SQ|SQC|GRBM|CPC|SPI)_[A-Za-z0-9_]+" | sort -u > "$OUT/counters.txt"
wc -l "$OUT/counters.txt"
i=0
for set in "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE" \
           "TA_BUSY_avr TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_ATOMIC_WITH_RET_REQ_sum" \
           "TCP_TCC_ATOMIC_WITHOUT_RET_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TA_TCP_STATE_READ_sum TCP_GATE_EN1_sum" \
           "TCC_REQ_sum TCC_ATOMIC_sum TCC_BUSY_avr TCC_TAG_STALL_sum" \
           "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_EA0_ATOMIC_sum TCC_HIT_sum TCC_MISS_sum" \
           "SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_INSTS_FLAT SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
    i=$((i + 1))
    timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d "$OUT/pmc$i" -- python tools/run_leg.py ${LEG:-c512} 2 < /dev/null > "$OUT/pmc$i.log" 2>&1
    tail -2 "$OUT/pmc$i.log" | cut -c1-200
done
python tools/pmc_summary.py "$OUT" > "$OUT/pmc_summary.json"
python - "$OUT" <<'PY'
import json, sys
d = json.load(open(sys.argv[1] + "/pmc_summary.json"))
for k, v in d.items():
    if "k_search" in k or "k_slot" in k:
        print(k)
        for c, x in sorted(v.items()):
            print("   %-40s launches %3d  mean %.4g  max %.4g" % (c, x["launches"], x["mean"], x["max"]))
PY
