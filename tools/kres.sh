#!/bin/bash
# Register / LDS / spill numbers of the search kernels, compile only (no GPU): tools/kres.sh [extra hipcc flags] 
# e.g. tools/kres.sh -DJD_SLOT_WPE=4
cd "$(dirname "$0")/.."
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off --offload-device-only -c \
  -Rpass-analysis=kernel-resource-usage -I include -I juicer_amd/csrc "$@" -o /tmp/jd_device_res.o juicer_amd/csrc/jd_device.hip 2>&1 | \
  python3 -c "
import re,sys
txt=sys.stdin.read()
blocks=re.split(r'remark: [^\n]*Function Name: ',txt)
for b in blocks[1:]:
    name=b.split('\n')[0].strip()
    if not re.search(r'k_resident|k_search|k_slot|gmm_kernel39|gmm_fast39', name): continue
    def g(k):
        m=re.search(k+r': (\d+)',b); return m.group(1) if m else '?'
    print('%-60s VGPR %s AGPR %s spillV %s spillS %s scratch %s LDS %s occ %s'%(name[:60],g('VGPRs'),g('AGPRs'),g('VGPRs Spill'),g('SGPRs Spill'),g('ScratchSize \[bytes/lane\]'),g('LDS Size \[bytes/block\]'),g('Occupancy \[waves/SIMD\]')))
"
