#!/bin/bash
# FETCH_SIZE / WRITE_SIZE calibration on known access counts (GPU box, through gpurun): tools/traffic_calib.sh
# -> gpurun_out/calib/{probe.log, pmc_fetch/, pmc_write/, calib.json}
set -u
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
OUT=gpurun_out/calib
rm -rf "$OUT"; mkdir -p "$OUT"
[ -x tools/traffic_probe ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/traffic_probe tools/traffic_probe.hip
rocprofv3 --list-avail > "$OUT/avail.txt" 2>&1 || rocprofv3 -L > "$OUT/avail.txt" 2>&1
tools/traffic_probe 8 > "$OUT/probe.log" 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- tools/traffic_probe 8 > "$OUT/pmc_fetch.log" 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -- tools/traffic_probe 8 > "$OUT/pmc_write.log" 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum --output-format csv -d "$OUT/pmc_rdreq" -- tools/traffic_probe 8 > "$OUT/pmc_rdreq.log" 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum --output-format csv -d "$OUT/pmc_wrreq" -- tools/traffic_probe 8 > "$OUT/pmc_wrreq.log" 2>&1
python tools/traffic_calib.py "$OUT" > "$OUT/calib.json"
cat "$OUT/calib.json"
