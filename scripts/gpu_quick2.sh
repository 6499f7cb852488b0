python scripts/bench_rollout_env.py balance 32768 100 | tail -1 | cut -c1-200
python scripts/bench_rollout_env.py transport 16384 100 | tail -1 | cut -c1-200
python scripts/bench_rollout_env.py navigation 8192 50 | tail -1 | cut -c1-200
python scripts/bench_bound.py balance 32768 | tail -1 | cut -c100-300
python scripts/bench_bound.py transport 16384 | tail -1 | cut -c100-300
ACTIONS=zero python scripts/bench_bound.py navigation 8192 | tail -1 | cut -c100-300
