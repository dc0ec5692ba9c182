#!/bin/bash
# register / spill statistics of the search kernels (cross-compiles, no GPU needed): tools/kstats.sh [extra hipcc flags]
set -e
out=${TMPDIR:-/tmp}/jd_kstats.s
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC "$@" -I "$(dirname "$0")/../include" -I "$(dirname "$0")/../juicer_amd/csrc" \
    --cuda-device-only -S "$(dirname "$0")/../juicer_amd/csrc/jd_device.hip" -o "$out" 2>/dev/null
python3 - "$out" <<'PY'
import re, sys
txt = open(sys.argv[1]).read()
for m in re.finditer(r'\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel', txt, re.S):
    name, body = m.group(1), m.group(2)
    if 'k_search' not in name and 'gmm_kernel39' not in name: continue
    g = lambda k: re.search(r'\.amdhsa_' + k + r' (\S+)', body)
    print(name, 'vgpr', g('next_free_vgpr').group(1), 'lds', g('group_segment_fixed_size').group(1), 'scratch', g('private_segment_fixed_size').group(1))
for m in re.finditer(r'^(_Z\S*k_search\S*):.*?\n; codeLenInByte = (\d+).*?; NumVgprs: (\d+).*?; ScratchSize: (\d+).*?; Occupancy: (\d+)', txt, re.S | re.M):
    pass
for name, sg, vg in re.findall(r'\.name:\s+(\S*k_search\S*)\n(?:.*\n)*?\s+\.sgpr_spill_count:\s+(\d+)\n(?:.*\n)*?\s+\.vgpr_spill_count:\s+(\d+)', txt):
    print(name, 'sgpr spills', sg, 'vgpr spills', vg)
PY
