# Round-2 call 2: the full GPU suite (reference suite on the HIP step, attach on cuda:0, two queues, football 131072),
# then A/B timings of this session's kernel changes and the VALU-by-phase counters of football.
mkdir -p gpurun_out/r02
rm -f gpurun_out/parity_allowance.jsonl
timeout 1200 python -m pytest tests -m gpu -q --durations=25 --timeout=300 -p no:cacheprovider > gpurun_out/r02/pytest_gpu_call2.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r02/pytest_gpu_call2.log
grep -E "^(FAILED|ERROR)|passed|failed|rc=" gpurun_out/r02/pytest_gpu_call2.log | cut -c1-300 | head -40
grep -E "^E  +(Assertion|.*Error)" gpurun_out/r02/pytest_gpu_call2.log | cut -c1-300 | head -20
S=scripts
{
for LIB in libvmas_hip.so libvmas_hip_x_base.so libvmas_hip_x_nols.so libvmas_hip_x_nofired.so libvmas_hip_x_noclamp.so; do
  export VMAS_HIP_LIB=$LIB
  QUEUES=1 python $S/bench_world.py football 131072 300
  QUEUES=1 python $S/bench_world.py balance 32768 2000
  QUEUES=1 python $S/bench_world.py balance 1048576 100
done
export VMAS_HIP_LIB=libvmas_hip.so
for Q in 1 2; do
  QUEUES=$Q python $S/bench_world.py balance 32768 3000
  QUEUES=$Q python $S/bench_world.py transport 16384 3000
  QUEUES=$Q python $S/bench_world.py navigation 65536 1000
  QUEUES=$Q python $S/bench_world.py navigation 8192 3000
  QUEUES=$Q python $S/bench_world.py football 16384 1000
done
unset VMAS_HIP_LIB
} 2>&1 | grep -v amdgpu.ids > gpurun_out/r02/ab_call2.log
cat gpurun_out/r02/ab_call2.log
python bench.py --no-cpu-baseline > gpurun_out/r02/bench_call2.json 2> gpurun_out/r02/bench_call2.err; tail -c 600 gpurun_out/r02/bench_call2.json
# VALU by phase (profile build): 0 full | 1 no items | 16 descriptors only | 32 broad phase only | 2 no integrate | 3 neither
cd /tmp && export TMPDIR=/tmp
for A in 0 1 16 32 2 3; do
  rm -rf /tmp/pv_$A
  VMAS_HIP_LIB=libvmas_hip_profile.so VMAS_ABLATE=$A QUEUES=1 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VALU_TRANS --output-format csv -d /tmp/pv_$A -o p -- python /root/repo/scripts/bench_world.py football 131072 60 > /tmp/pv_$A.log 2>&1
  python - $A <<'P'
import csv, glob, sys, collections
a = sys.argv[1]
acc = collections.defaultdict(list)
for f in glob.glob(f"/tmp/pv_{a}/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "step_kernel<0, 0" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
m = {k: sum(v) / len(v) for k, v in acc.items()}
w = m.get("SQ_WAVES", 1)
print(f"football131072 ablate={a} per wave: " + ", ".join(f"{k[3:]} {v / w:.0f}" for k, v in sorted(m.items()) if k != "SQ_WAVES"))
P
done > /root/repo/gpurun_out/r02/football_valu_by_phase.txt 2>&1
cat /root/repo/gpurun_out/r02/football_valu_by_phase.txt
