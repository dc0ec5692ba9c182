export TMPDIR=/tmp
OUT=gpurun_out/gmmpmc; rm -rf $OUT; mkdir -p $OUT
ONE="python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extra-legs"
i=0
for set in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE" "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_ANY SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_LDS" "SQ_INST_CYCLES_VMEM SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAVES SQ_INSTS_VMEM_RD SQ_THREAD_CYCLES_VALU"; do
  i=$((i+1))
  timeout 600 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/p$i -- $ONE < /dev/null > $OUT/p$i.log 2>&1
  f=$(find $OUT/p$i -name '*counter_collection.csv' | head -1)
  python - "$f" <<'PY'
import csv,sys,collections
acc=collections.defaultdict(float)
for r in csv.DictReader(open(sys.argv[1])):
    if 'gmm' in r['Kernel_Name']:
        acc[r['Counter_Name']]+=float(r['Counter_Value'])
print(dict(acc))
PY
done
