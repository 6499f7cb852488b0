# PMC counters of the fused env.step kernels (separate pass from the kernel trace)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
S=${1:-balance}; B=${2:-32768}
rm -rf /tmp/pmc_env
ONLY=fused-eager rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES --output-format csv -d /tmp/pmc_env -o env -- python $R/scripts/bench_env.py $S $B > /tmp/pmc_env.log 2>&1
python - <<'PY'
import csv, glob, collections
fs = glob.glob('/tmp/pmc_env/**/*counter_collection.csv', recursive=True)
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(fs[0])):
    k = r['Kernel_Name'].replace('(anonymous namespace)::', '').split('(')[0][-40:]
    acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
for k, d in acc.items():
    if 'kernel' not in k or 'at::' in k: continue
    print(k, {c: round(sum(v) / len(v)) for c, v in d.items()})
PY
