for H in 0 20 150 600; do HOLD=$H python scripts/trace_compact.py football 131072 2>&1 | tail -11; done
python scripts/bench_rollout_env.py football 131072 50 | tail -1 | cut -c1-160
