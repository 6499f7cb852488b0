timeout 900 python -m pytest tests/test_hip_parity.py tests/test_broad_phase_full_size_gpu.py -q --timeout=600 -p no:cacheprovider -m gpu -x 2>&1 | tail -3
export VMAS_HIP_LIB=libvmas_hip_profile.so
for B in 65536 131072 262144 1048576; do
  for P in 0 1; do
    for Q in 1 2; do
      VMAS_PERSISTENT=$P QUEUES=$Q python scripts/bench_world.py balance $B 300 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('balance', d['num_envs'], 'persistent', $P, 'queues', d['queues'], 'us', d['world_step_us'], 'frac', round(384*d['num_envs']/d['world_step_us']/1e3/8000,3))"
    done
  done
done
for B in 131072 1048576; do
  for P in 0 1; do
    VMAS_PERSISTENT=$P QUEUES=1 python scripts/bench_world.py transport $B 300 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('transport', d['num_envs'], 'persistent', $P, 'us', d['world_step_us'])"
  done
done
