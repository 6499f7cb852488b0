mkdir -p gpurun_out/r3q
export NAV_TILES=64
for B in 65536 8192; do
  for ACT in fixed zero; do
    ACTIONS=$ACT python scripts/bench_bound.py navigation $B | tail -1
    for AB in 0 1 2 3 4; do
      VMAS_HIP_LIB=libvmas_hip_profile.so VMAS_ENV_ABLATE=$AB ACTIONS=$ACT python scripts/bench_bound.py navigation $B | tail -1
    done
  done
done
cd /tmp && export TMPDIR=/tmp
ACTIONS=zero rocprofv3 --kernel-trace --stats -d /tmp/prof_nav -o nav -- python /root/repo/scripts/bench_bound.py navigation 65536 > /dev/null 2>&1
python - <<'P'
import csv, glob
for f in glob.glob("/tmp/prof_nav/**/*kernel_stats.csv", recursive=True):
    for r in list(csv.DictReader(open(f)))[:6]:
        print(r["Name"][:90], r["Calls"], r["AverageNs"], r["Percentage"])
P
