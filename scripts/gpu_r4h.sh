# Round 4, eighth GPU call: same-box A/B of two libraries - libvmas_hip_head.so (the previous commit, built from a worktree
# of it) against libvmas_hip.so (load-phase changes + football's observations as contiguous runs) - interleaved, because
# boxes differ by up to 15 % on the large-batch kernels (r04d vs r04g); then the tests on the new library
TAG=r04h
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd $R
S=$R/scripts
AB=$OUT/${TAG}_ab_previous_commit_vs_this.jsonl
: > $AB
for ROUND in 1 2; do
  for LIB in libvmas_hip_head.so libvmas_hip.so; do
    export VMAS_HIP_LIB=$LIB
    { COMPACT=1 FORCES=random QUEUES=1 python $S/bench_world.py football 131072 300
      COMPACT=1 FORCES=random QUEUES=2 python $S/bench_world.py football 131072 300
      COMPACT=1 FORCES=random python $S/bench_world.py football 16384 300
      python $S/bench_bound.py football 131072
      REPS=5 python $S/bench_rollout_env.py football 131072 50
      REPS=5 python $S/bench_rollout_env.py football 16384 50
      python $S/bench_bound.py balance 32768
      python $S/bench_rollout_env.py balance 32768 100
      FORCES=random python $S/bench_world.py balance 32768 2000
    } 2>&1 | grep "^{" | sed "s/^{/{\"ab_library\": \"$LIB\", \"round\": $ROUND, /" >> $AB
  done
done
unset VMAS_HIP_LIB
cat $AB | cut -c1-420
# the shared observation array in the latency regime too? (profile build: the knob forces the form)
{ for ROWS in 0 28 64; do VMAS_HIP_LIB=libvmas_hip_profile.so VMAS_FOOTBALL_STAGE_ROWS=$ROWS REPS=5 python $S/bench_rollout_env.py football 16384 50 2>&1 | grep "^{" | sed "s/^{/{\"stage_rows\": $ROWS, /"; done
  for ROWS in 0 16 28; do VMAS_HIP_LIB=libvmas_hip_profile.so VMAS_FOOTBALL_STAGE_ROWS=$ROWS REPS=5 python $S/bench_rollout_env.py football 131072 50 2>&1 | grep "^{" | sed "s/^{/{\"stage_rows\": $ROWS, /"; done
  VMAS_HIP_LIB=libvmas_hip_profile.so VMAS_FOOTBALL_NO_STAGE=1 REPS=5 python $S/bench_rollout_env.py football 131072 50 2>&1 | grep "^{" | sed "s/^{/{\"stage_rows\": \"none\", /"
} > $OUT/${TAG}_football_observation_staging_ab.jsonl; cat $OUT/${TAG}_football_observation_staging_ab.jsonl | cut -c1-300
timeout 1500 python -m pytest tests -m gpu -q --timeout=600 -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log
grep -E "^(FAILED|ERROR)|passed|failed|rc=|needed it|fixtures with any" $OUT/pytest_gpu.log | cut -c1-300 | head -20
grep -E "^E  +(Assertion|.*Error)" $OUT/pytest_gpu.log | cut -c1-300 | head -20
python bench.py --config football --no-cpu-baseline --no-attached > $OUT/${TAG}_bench_line_football.json 2>> $OUT/bench.err; cut -c1-1200 $OUT/${TAG}_bench_line_football.json
