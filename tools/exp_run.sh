#!/bin/bash
# development: one line per (JD_EXP variant, workload) -> gpurun_out/exp.log    usage: tools/exp_run.sh "0 1 2" "c2 clg c3 north"
cd "$(dirname "$0")/.." || exit 1
export JD_BENCH_NO_LAZY=1
for v in $1; do
  for leg in $2; do
    if [ "$leg" = c2 ]; then
      r=$(JD_EXP=$v python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extra-legs 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['roofline']['search_ms_per_step'], d['roofline']['gmm']['ms_per_step'])")
    else
      r=$(JD_EXP=$v python tools/run_leg.py $leg 3 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print(d['ms_per_step'], d['search_ms'], d['gmm_ms'], d['hyps_found'], d['per_stream_frame'])")
    fi
    echo "exp=$v $leg: $r"
  done
done
