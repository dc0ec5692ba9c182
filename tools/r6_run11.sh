cp build_ab/lib_fine.so juicer_amd/libjuicer_amd.so
python tools/slot_trace.py --slots 256 --depth 9 --steps 30 2>&1 | grep -v amdgpu.ids | tail -16
