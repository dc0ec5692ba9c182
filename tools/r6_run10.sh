for v in nodefer defer nodefer defer; do
  cp build_ab/lib_$v.so juicer_amd/libjuicer_amd.so
  echo "== $v"
  python tools/slot_trace.py --slots 256 --depth 9 --steps 30 2>&1 | grep -v amdgpu.ids | grep "^slots\|phase A\|phase X\|sum"
done
cp build_ab/lib_defer.so juicer_amd/libjuicer_amd.so
python -m pytest tests/test_gpu_slot.py tests/test_gpu_fullsize.py tests/test_gpu_parity.py -x -q -m gpu -k "slot or resident or pipeline or fullsize or batch" 2>&1 | tail -5
