export JD_DEV=1
for cfg in "0 256 9" "5 320 11" "6 384 13" "6 384 24" "7 448 15" "4 256 9"; do
  set -- $cfg
  if [ "$1" = "0" ]; then unset JD_SLOT_KEEP_SE; else export JD_SLOT_KEEP_SE=$1; fi
  echo "== keep_se $1 slots $2 depth $3 fast"
  JD_VERBOSE=1 python tools/slot_trace.py --slots $2 --depth $3 --steps 30 --scoring fast 2>&1 | grep -v "amdgpu.ids\|^arenas\|arc order\|k_search:\|^  " | tail -12
done
