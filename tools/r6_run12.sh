for v in base prio base prio; do
  cp build_ab/lib_$v.so juicer_amd/libjuicer_amd.so
  echo "== $v"
  python tools/slot_trace.py --slots 256 --depth 9 --steps 30 2>&1 | grep -v amdgpu.ids | grep "^slots\|phase A\|phase X\|sum"
done
