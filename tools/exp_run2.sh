#!/bin/bash
# development: tools/exp_run2.sh "<env assignments>;<env assignments>;..." "c2 clg c3 north"
cd "$(dirname "$0")/.." || exit 1
export JD_BENCH_NO_LAZY=1
IFS=';' read -ra VARS <<< "$1"
for v in "${VARS[@]}"; do
  for leg in $2; do
    if [ "$leg" = c2 ]; then
      r=$(env $v python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extra-legs 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['roofline']['search_ms_per_step'], d['roofline']['gmm']['ms_per_step'])")
    else
      r=$(env $v JD_VERBOSE=1 python tools/run_leg.py $leg 3 2>/tmp/err.log | python -c "import sys,json; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print(d['ms_per_step'], d['search_ms'], d['gmm_ms'], d['hyps_found'], d['per_stream_frame']['tot_arcs_visited'])")
      x=$(grep -c "XCD-local" /tmp/err.log); y=$(grep -c "k_search:" /tmp/err.log)
      r="$r xl_launches=$x/$y"
    fi
    echo "[$v] $leg: $r"
  done
done
