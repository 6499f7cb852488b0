export VMAS_HIP_LIB=libvmas_hip_profile.so
for NS in "" 1; do
  echo "NO_STAGE=$NS"
  VMAS_FOOTBALL_NO_STAGE=$NS REPS=5 python scripts/bench_rollout_env.py football 131072 50 | tail -1
  VMAS_FOOTBALL_NO_STAGE=$NS REPS=20 python scripts/bench_rollout_env.py football 131072 10 | tail -1
  VMAS_FOOTBALL_NO_STAGE=$NS python scripts/bench_rollout_env.py football 16384 50 | tail -1
done
