#!/bin/bash
# development: library variants built beforehand (tools/variants/<name>.so), each timed with the next table scored ahead
# (tools/pf_ab.py), alternately on ONE box     usage: tools/ab_pf.sh "A_base B_x A_base B_x" "c2 north" [rounds]
cd "$(dirname "$0")/.." || exit 1
for v in $1; do
  cp tools/variants/$v.so juicer_amd/libjuicer_amd.so
  for leg in $2; do
    r=$(python tools/pf_ab.py $leg ${3:-2} ahead= serial=JD_NOTHING:0 2>/dev/null | grep -o '"variant.*median_ms[^,]*' | tr '\n' ' ')
    echo "[$v] $leg: $r"
  done
done
