#!/bin/bash
# development: library variants built beforehand (tools/variants/<name>.so) timed alternately on ONE box (boxes differ by
# several per cent)    usage: tools/ab_run.sh "A_packed B_ahead A_packed B_ahead" "c2 north"
cd "$(dirname "$0")/.." || exit 1
for v in $1; do
  cp tools/variants/$v.so juicer_amd/libjuicer_amd.so
  echo "variant $v"
  tools/exp_run.sh "0" "$2"
done
