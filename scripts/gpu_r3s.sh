export NAV_TILES=64
for B in 65536 8192; do
  for AB in 4 12 28; do
    VMAS_HIP_LIB=libvmas_hip_profile.so VMAS_ENV_ABLATE=$AB ACTIONS=zero timeout 120 python scripts/bench_bound.py navigation $B | tail -1
  done
  QUEUES=1 python scripts/bench_world.py navigation $B | tail -1
done
for AB in 0 4 12; do
VMAS_HIP_LIB=libvmas_hip_profile.so VMAS_ENV_ABLATE=$AB timeout 120 python scripts/bench_bound.py balance 32768 | tail -1
done
