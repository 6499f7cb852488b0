# Phase ablations of the step kernel, time and instruction counts, for any native world:
#   bash scripts/gpu_ablate.sh football 131072 [steps]
# The VMAS_ABLATE knob exists only in the -DVMAS_PROFILE build (libvmas_hip_profile.so, built beside the product
# library: VMAS_HIPCC_EXTRA=-DVMAS_PROFILE VMAS_LIB_OUT=libvmas_hip_profile.so bash csrc/build.sh):
#   0 full | 1 no items | 16 descriptors only | 32 broad phase only | 2 no integrate | 3 neither | 8 no trig
W=${1:-football}; B=${2:-131072}; N=${3:-200}
R=${GRAFT_REPO_ROOT:-$(pwd)}
export VMAS_HIP_LIB=libvmas_hip_profile.so QUEUES=1 SPEC=0   # (the phases of the INTERPRETER: step_kernel)
cd /tmp && export TMPDIR=/tmp
for A in 0 1 16 32 2 3; do
  T=$(VMAS_ABLATE=$A python $R/scripts/bench_world.py $W $B $N 2>/dev/null | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['world_step_us'])")
  rm -rf /tmp/pv_$A
  VMAS_ABLATE=$A rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d /tmp/pv_$A -o p -- python $R/scripts/bench_world.py $W $B 40 > /tmp/pv_$A.log 2>&1
  python - $A $W $B $T <<'P'
import csv, glob, sys, collections
a, w, b, t = sys.argv[1:5]
acc = collections.defaultdict(list)
for f in glob.glob(f"/tmp/pv_{a}/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "step_kernel" in r["Kernel_Name"] and "NoEnv" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
m = {k: sum(v) / len(v) for k, v in acc.items()}
wv = m.get("SQ_WAVES", 1)
print(f"{w} {b} ablate={a}: {t} us/step; per wave: " + ", ".join(f"{k[9:]} {v / wv:.0f}" for k, v in sorted(m.items()) if k != "SQ_WAVES"))
P
done
