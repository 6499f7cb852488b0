# (the VMAS_ABLATE / VMAS_ENV_ABLATE knobs only exist in -DVMAS_PROFILE builds; the product library is rebuilt at the end)
VMAS_HIPCC_EXTRA=-DVMAS_PROFILE bash vectorizedmultiagentsimulator_amd/csrc/build.sh > /dev/null 2>&1
# kernel duration of the one-launch balance step with pieces of the prologue / epilogue switched off
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for A in ${ABL:-0 1 2 3 4 8 12}; do
  rm -rf /tmp/prof_env
  VMAS_ENV_ABLATE=$A ONLY=fused-eager rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_env -o env -- python $R/scripts/bench_env.py ${1:-balance} ${2:-32768} > /tmp/prof_env.log 2>&1
  f=$(find /tmp/prof_env -name "*kernel_stats.csv" | head -1)
  python - "$A" "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[2])):
    if "step_kernel" in r["Name"]:
        print(f"ablate {sys.argv[1]}: {float(r['AverageNs'])/1000:.2f} us ({r['Calls']} calls)")
PY
done
bash vectorizedmultiagentsimulator_amd/csrc/build.sh > /dev/null 2>&1
