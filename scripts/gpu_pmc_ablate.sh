# (the VMAS_ABLATE / VMAS_ENV_ABLATE knobs only exist in -DVMAS_PROFILE builds; the product library is rebuilt at the end)
VMAS_HIPCC_EXTRA=-DVMAS_PROFILE bash vectorizedmultiagentsimulator_amd/csrc/build.sh > /dev/null 2>&1
# dynamic instruction counts per wave of step_kernel under the VMAS_ABLATE profiling toggles
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
for A in ${ABL:-0 1 16 32 2 3 15}; do
  OUT=/tmp/pmc_abl_$A; rm -rf $OUT
  VMAS_ABLATE=$A rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VALU_TRANS SQ_WAVES --output-format csv -d $OUT -o b -- python bench.py --no-cpu-baseline --no-fused --steps 200 --warmup 20 ${BENCH_ARGS:-} > /dev/null 2>&1
  python - $A $OUT <<'PY'
import csv, glob, collections, sys
fs=glob.glob(sys.argv[2]+'/**/*counter_collection.csv', recursive=True)
acc=collections.defaultdict(list)
for r in csv.DictReader(open(fs[0])):
    if 'step_kernel' in r['Kernel_Name']: acc[r['Counter_Name']].append(float(r['Counter_Value']))
w=sum(acc['SQ_WAVES'])/len(acc['SQ_WAVES'])
print('ablate', sys.argv[1], ' '.join('%s %.0f'%(k.replace('SQ_INSTS_',''), sum(v)/len(v)/w) for k,v in sorted(acc.items()) if k!='SQ_WAVES'), 'waves %d'%w)
PY
done
bash vectorizedmultiagentsimulator_amd/csrc/build.sh > /dev/null 2>&1
