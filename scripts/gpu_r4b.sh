export VMAS_HIP_LIB=libvmas_hip_profile.so
VMAS_PERSISTENT=1 QUEUES=1 python scripts/bench_world.py balance 1048576 100 | tail -1
VMAS_PERSISTENT=0 QUEUES=1 python scripts/bench_world.py balance 1048576 100 | tail -1
unset VMAS_HIP_LIB
QUEUES=1 python scripts/bench_world.py balance 1048576 100 | tail -1
QUEUES=2 python scripts/bench_world.py balance 1048576 100 | tail -1
QUEUES=1 python scripts/bench_world.py balance 262144 100 | tail -1
QUEUES=1 python scripts/bench_world.py balance 131072 100 | tail -1
