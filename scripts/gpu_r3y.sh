timeout 1200 python -m pytest tests/test_env_fused_gpu.py tests/test_compact_gpu.py tests/test_env_gpu.py -q --timeout=600 -p no:cacheprovider -m gpu -x -k football 2>&1 | tail -4
python scripts/bench_rollout_env.py football 131072 50 | tail -1
python scripts/bench_rollout_env.py football 16384 50 | tail -1
python scripts/bench_rollout_env.py football 32768 50 | tail -1
ONLY=fused-eager python scripts/bench_env.py football 131072 | tail -1
