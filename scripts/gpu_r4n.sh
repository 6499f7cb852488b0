# Round 4, thirteenth GPU call: the compacted kernel's load phase driven by one planner word per entity (no mask / popcount arithmetic,
# tail lanes read a clamped column instead of being predicated) against the build of the r04l evidence (libvmas_hip_prev.so), same box; tests
TAG=r04n
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd $R
S=$R/scripts
timeout 1500 python -m pytest tests -m gpu -q --timeout=600 -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log
grep -E "^(FAILED|ERROR)|passed|failed|rc=|needed it|fixtures with any" $OUT/pytest_gpu.log | cut -c1-300 | head -20
grep -E "^E  +(Assertion|.*Error)" $OUT/pytest_gpu.log | cut -c1-300 | head -20
AB=$OUT/${TAG}_ab_previous_commit_vs_this.jsonl
: > $AB
for LIB in libvmas_hip_prev.so libvmas_hip.so libvmas_hip_prev.so libvmas_hip.so; do
  export VMAS_HIP_LIB=$LIB
  { COMPACT=1 FORCES=random QUEUES=1 python $S/bench_world.py football 131072 300
    COMPACT=1 FORCES=random QUEUES=2 python $S/bench_world.py football 131072 300
    COMPACT=1 FORCES=random python $S/bench_world.py football 16384 300
    COMPACT=1 FORCES=random python $S/bench_world.py football 8192 300
    [ $LIB = libvmas_hip.so ] && REPS=5 python $S/bench_rollout_env.py football 16384 50
    [ $LIB = libvmas_hip.so ] && REPS=5 python $S/bench_rollout_env.py football 131072 50
  } 2>&1 | grep "^{" | sed "s/^{/{\"ab_library\": \"$LIB\", /" >> $AB
done
unset VMAS_HIP_LIB
python - <<P
import json
for l in open("$AB"):
    r = json.loads(l)
    print(r["ab_library"].ljust(20), r["num_envs"], r.get("queues"), {k: v for k, v in r.items() if k.endswith("_us") or "us_per_step" in k})
P
python $S/trace_compact.py football 16384 2>&1 | grep -v amdgpu > $OUT/${TAG}_football16384_compact_phase_trace.txt; tail -n 32 $OUT/${TAG}_football16384_compact_phase_trace.txt
