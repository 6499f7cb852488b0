echo "=== streams balance 32768"; python scripts/exp_streams.py balance 32768 2000
echo "=== football ablation (profile build): 0 full | 1 no items | 16 descriptors only | 32 broad phase only | 2 no integrate | 3 neither | 8 no trig"
export VMAS_HIP_LIB=libvmas_hip_profile.so
for A in 0 1 16 32 2 3 8; do VMAS_ABLATE=$A python scripts/bench_world.py football 131072 200; done
echo "=== navigation ablation"
for A in 0 1 16 32 2 3; do VMAS_ABLATE=$A python scripts/bench_world.py navigation 65536 200; done
echo "=== balance 1M ablation"
for A in 0 1 16 32 2 3; do VMAS_ABLATE=$A python scripts/bench_world.py balance 1048576 100; done
