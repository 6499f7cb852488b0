export ACTIONS=zero
python scripts/bench_bound.py navigation 16384 | tail -1 | cut -c100-260
python scripts/bench_bound.py navigation 16384 | tail -1 | cut -c100-260
OUT=gpurun_out/r4h; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -o env -- python scripts/bench_bound.py navigation 16384 > $OUT/stdout.log 2>&1
tail -1 $OUT/stdout.log | cut -c100-260
f=$(find $OUT/trace -name "*kernel_trace.csv" | head -1)
python - "$f" <<'P'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "step_kernel" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-1500:]
d = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows]
g = [int(rows[i + 1]["Start_Timestamp"]) - int(rows[i]["End_Timestamp"]) for i in range(len(rows) - 1)]
d.sort(); g.sort()
print("kernel us median", d[len(d)//2] / 1e3, "gap us median", g[len(g)//2] / 1e3, "p90", g[int(len(g)*0.9)] / 1e3, "queues", set(r.get("Queue_Id") for r in rows))
P
rm -rf $OUT/trace
HIP_FORCE_DEV_KERNARG=1 python scripts/bench_bound.py navigation 16384 | tail -1 | cut -c100-260
GPU_MAX_HW_QUEUES=1 python scripts/bench_bound.py navigation 16384 | tail -1 | cut -c100-260
