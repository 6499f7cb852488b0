export NAV_TILES=64
prof() { # name, command...
  OUT=gpurun_out/r3t_$1; rm -rf $OUT; mkdir -p $OUT; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o env -- "$@" > $OUT/stdout.log 2>&1
  f=$(find $OUT/trace -name "*kernel_trace.csv" | head -1)
  python - "$f" <<'P'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(list)
for r in rows:
    if "step_kernel" in r["Kernel_Name"] or "collision" in r["Kernel_Name"]:
        agg[(r["Kernel_Name"][:80], r.get("LDS_Block_Size"), r.get("VGPR_Count"), r.get("Accum_VGPR_Count"), r.get("Scratch_Size"), r.get("Workgroup_Size"), r.get("Grid_Size"))].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k, v in agg.items():
    v.sort(); print(k, "n", len(v), "median_us", v[len(v)//2] / 1e3)
P
  rm -rf $OUT/trace
}
export ACTIONS=zero
VMAS_HIP_LIB=libvmas_hip_profile.so VMAS_ENV_ABLATE=28 prof a28 python scripts/bench_bound.py navigation 65536
VMAS_HIP_LIB=libvmas_hip_profile.so VMAS_ENV_ABLATE=4 prof a4 python scripts/bench_bound.py navigation 65536
prof full python scripts/bench_bound.py navigation 65536
QUEUES=1 prof world python scripts/bench_world.py navigation 65536
prof world2q python scripts/bench_world.py navigation 65536
