# Round 4, ninth GPU call: football's Environment.step as two launches per step (step kernel + stand-alone post-step with
# contiguous observation runs) against the one-launch form, by batch size (profile build: VMAS_FOOTBALL_SPLIT forces the form);
# the previous commit's library beside this one on the same box (the compacted kernel's loads reverted: the regression of
# r04g/r04h must be gone); tests
TAG=r04i
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd $R
S=$R/scripts
timeout 1500 python -m pytest tests -m gpu -q --timeout=600 -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log
grep -E "^(FAILED|ERROR)|passed|failed|rc=|needed it|fixtures with any" $OUT/pytest_gpu.log | cut -c1-300 | head -20
grep -E "^E  +(Assertion|.*Error)" $OUT/pytest_gpu.log | cut -c1-300 | head -20
{ for B in 16384 32768 65536 131072; do for SPLIT in 0 1; do
    VMAS_HIP_LIB=libvmas_hip_profile.so VMAS_FOOTBALL_SPLIT=$SPLIT REPS=5 python $S/bench_rollout_env.py football $B 50 2>&1 | grep "^{" | sed "s/^{/{\"two_launches_per_step\": $SPLIT, /"
  done; done; } > $OUT/${TAG}_football_env_step_one_launch_vs_two.jsonl; cut -c1-330 $OUT/${TAG}_football_env_step_one_launch_vs_two.jsonl
AB=$OUT/${TAG}_ab_previous_commit_vs_this.jsonl
: > $AB
for LIB in libvmas_hip_head.so libvmas_hip.so libvmas_hip_head.so libvmas_hip.so; do
  export VMAS_HIP_LIB=$LIB
  { COMPACT=1 FORCES=random QUEUES=1 python $S/bench_world.py football 131072 300
    COMPACT=1 FORCES=random QUEUES=2 python $S/bench_world.py football 131072 300
    COMPACT=1 FORCES=random python $S/bench_world.py football 16384 300
    REPS=5 python $S/bench_rollout_env.py football 131072 50
    python $S/bench_bound.py balance 32768
    FORCES=random python $S/bench_world.py balance 32768 2000
  } 2>&1 | grep "^{" | sed "s/^{/{\"ab_library\": \"$LIB\", /" >> $AB
done
unset VMAS_HIP_LIB
cut -c1-400 $AB
for LIB in libvmas_hip_head.so libvmas_hip.so; do
  VMAS_HIP_LIB=$LIB python bench.py --config football --no-cpu-baseline --no-attached > $OUT/${TAG}_bench_line_football_${LIB%.so}.json 2>> $OUT/bench.err
  python - <<P
import json
d = json.loads(open("$OUT/${TAG}_bench_line_football_${LIB%.so}.json").read().strip().splitlines()[-1])
print("$LIB", {k: d.get(k) for k in ("value", "ms_per_step")}, {k: (v if not isinstance(v, dict) else {kk: vv for kk, vv in v.items() if "us" in kk or "value" in kk}) for k, v in d.items() if k in ("environment_step", "persistent_rollout")})
P
done
python $S/trace_compact.py football 16384 2>&1 | grep -v amdgpu > $OUT/${TAG}_football16384_compact_phase_trace.txt; tail -n 16 $OUT/${TAG}_football16384_compact_phase_trace.txt
