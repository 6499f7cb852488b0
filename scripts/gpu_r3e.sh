mkdir -p gpurun_out/r3i
timeout 900 python -m pytest tests/test_compact_gpu.py tests/test_env_fused_gpu.py -k "compact or football" -q --timeout=300 -p no:cacheprovider > gpurun_out/r3i/pytest_compact.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3i/pytest_compact.log
grep -E "^(FAILED|ERROR)|passed|failed|rc=" gpurun_out/r3i/pytest_compact.log | cut -c1-300 | head -20
grep -E "^E  +" gpurun_out/r3i/pytest_compact.log | cut -c1-300 | head -20
for W in "football 1024" "football 131072"; do python scripts/trace_compact.py $W 2>&1 | tail -11; done | tee gpurun_out/r3i/trace_compact.txt
{
for F in random fixed; do for W in "football 131072" "football 16384"; do
  for CP in 0 1; do FORCES=$F COMPACT=$CP QUEUES=1 python scripts/bench_world.py $W 100; done
  FORCES=$F COMPACT=1 QUEUES=2 python scripts/bench_world.py $W 100
done; done
ONLY=fused-eager python scripts/bench_env.py football 131072
ONLY=fused-eager python scripts/bench_env.py football 16384
} 2>&1 | grep "^{\|Error\|error" | cut -c1-600 > gpurun_out/r3i/rates.jsonl
cat gpurun_out/r3i/rates.jsonl
cd /tmp && export TMPDIR=/tmp
FORCES=random COMPACT=1 QUEUES=1 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU --output-format csv -d /tmp/pm -o p -- python $GRAFT_REPO_ROOT/scripts/bench_world.py football 131072 30 > /tmp/pm.log 2>&1
python - <<'P'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("/tmp/pm/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "step_kernel" in r["Kernel_Name"]:
            acc[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, c in acc.items():
    m = {n: sum(v) / len(v) for n, v in c.items()}
    wv = m.get("SQ_WAVES", 1)
    print(k, {n: round(v / wv, 1) for n, v in m.items()})
P
