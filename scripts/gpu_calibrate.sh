# FETCH_SIZE / WRITE_SIZE calibration in the step kernel's access pattern (separate PMC passes)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for n in 1572864 134217728; do   # 6.3 MB (cache resident between dispatches, like the 32768-env state) and 512 MB (past L3)
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/cal
    rocprofv3 --pmc $c --output-format csv -d /tmp/cal -o cal -- python $R/scripts/calibrate_traffic.py $n 20 > /tmp/cal.log 2>&1
    python - "$n" "$c" <<'PY'
import csv, glob, sys
n, c = int(sys.argv[1]), sys.argv[2]
f = glob.glob('/tmp/cal/**/*counter_collection.csv', recursive=True)[0]
v = [float(r['Counter_Value']) for r in csv.DictReader(open(f)) if 'math_kernel' in r['Kernel_Name'] and r['Counter_Name'] == c]
v = v[2:]  # first dispatches: cold
m = sum(v) / len(v)
print(f"n={n} ({4*n/2**20:.1f} MiB each way) {c}: {m:.1f} KiB per dispatch = {m*1024/(4*n):.4f} of the known bytes")
PY
  done
done
