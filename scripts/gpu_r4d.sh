timeout 900 python -m pytest tests/test_hip_parity.py tests/test_env_fused_gpu.py -q --timeout=600 -p no:cacheprovider -m gpu -x 2>&1 | tail -3
p() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['num_envs'], 'lanes', d['lanes'], 'queues', d['queues'], 'spec', d['specialized'], 'us', d['world_step_us'], 'frac', round(384*d['num_envs']/d['world_step_us']/1e3/8000,3))"; }
for B in 65536 131072 262144 1048576; do for L in 4 8; do for Q in 1 2; do LANES=$L QUEUES=$Q python scripts/bench_world.py balance $B 200 | tail -1 | p balance; done; done; done
python scripts/bench_bound.py balance 65536 | tail -1
LANES=8 python scripts/bench_bound.py balance 65536 | tail -1
python bench.py --gpus 1 --steps 2000 --warmup 200 --no-cpu-baseline 2>&1 | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bench', d['value'], d['ms_per_step'], d['roofline'])"
