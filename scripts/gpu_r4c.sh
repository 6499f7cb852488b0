p() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['num_envs'], 'lanes', d['lanes'], 'queues', d['queues'], 'spec', d['specialized'], 'us', d['world_step_us'])"; }
for B in 65536 131072 1048576; do for L in 4 8; do LANES=$L QUEUES=1 python scripts/bench_world.py balance $B 200 | tail -1 | p balance; done; done
for B in 32768 131072 1048576; do for L in 8 16; do LANES=$L QUEUES=1 python scripts/bench_world.py transport $B 200 | tail -1 | p transport; done; done
export VMAS_HIP_LIB=libvmas_hip_profile.so
for B in 131072 262144 1048576; do for P in 0 1; do for Q in 1 2; do
  echo -n "persistent=$P "; VMAS_PERSISTENT=$P QUEUES=$Q python scripts/bench_world.py balance $B 200 | tail -1 | p balance
done; done; done
for P in 0 1; do echo -n "persistent=$P "; VMAS_PERSISTENT=$P QUEUES=1 python scripts/bench_world.py transport 1048576 200 | tail -1 | p transport; done
