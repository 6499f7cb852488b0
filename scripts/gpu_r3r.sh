export NAV_TILES=64
timeout 120 python scripts/bench_bound.py balance 32768 | tail -1
ACTIONS=zero timeout 120 python scripts/bench_bound.py navigation 65536 | tail -1
ACTIONS=zero timeout 120 python scripts/bench_bound.py navigation 8192 | tail -1
timeout 120 python scripts/bench_bound.py transport 32768 | tail -1
timeout 900 python -m pytest tests/test_env_fused_gpu.py -q --timeout=600 -p no:cacheprovider -x 2>&1 | tail -3
