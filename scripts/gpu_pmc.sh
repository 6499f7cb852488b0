OUT=gpurun_out/pmc; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
CMD="python bench.py --no-cpu-baseline --steps 300 --warmup 50 ${BENCH_ARGS:-}"
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD --output-format csv -d $OUT/p1 -o b -- $CMD > /dev/null 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU --output-format csv -d $OUT/p2 -o b -- $CMD > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_VALU_TRANS SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_BRANCH SQ_INST_LEVEL_LDS SQ_INST_LEVEL_SMEM SQ_INST_LEVEL_VMEM SQ_THREAD_CYCLES_VALU SQ_INSTS_SENDMSG --output-format csv -d $OUT/p3 -o b -- $CMD > /dev/null 2>&1
python - <<'PY'
import csv, glob, collections
for d in ('p1','p2','p3'):
    fs=glob.glob(f'gpurun_out/pmc/{d}/**/*counter_collection.csv', recursive=True)
    if not fs: print(d,'no csv'); continue
    acc=collections.defaultdict(list)
    for r in csv.DictReader(open(fs[0])):
        if 'step_kernel' in r['Kernel_Name']: acc[r['Counter_Name']].append(float(r['Counter_Value']))
    for k,v in acc.items(): print(d, k, '%.5g'%(sum(v)/len(v)), 'per wave %.1f'%(sum(v)/len(v)/4096))
PY
find $OUT -name "*.csv" -size +2M -delete
