export JD_DEV=1
for cfg in "6 384 13 6144" "6 384 13 16384" "6 384 13 49152" "5 320 11 16384" "7 448 15 49152"; do
  set -- $cfg
  export JD_SLOT_KEEP_SE=$1 JD_PIPE_PIECE=$4
  echo "== keep_se $1 slots $2 depth $3 piece $4 fast"
  python tools/slot_trace.py --slots $2 --depth $3 --steps 30 --scoring fast --no-trace 2>&1 | grep "^slots"
done
