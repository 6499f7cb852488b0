export NAV_TILES=64
timeout 900 python -m pytest tests/test_env_fused_gpu.py tests/test_env_gpu.py tests/test_scenarios_vs_reference.py -q --timeout=600 -p no:cacheprovider -m gpu -x 2>&1 | tail -5
ACTIONS=zero timeout 120 python scripts/bench_bound.py navigation 65536 | tail -1
ACTIONS=zero timeout 120 python scripts/bench_bound.py navigation 8192 | tail -1
timeout 120 python scripts/bench_bound.py balance 32768 | tail -1
timeout 120 python scripts/bench_bound.py transport 32768 | tail -1
VMAS_TRACE=2 python scripts/trace_nav.py 65536 2>&1 | tail -18
VMAS_TRACE=2 python scripts/trace_nav.py 8192 2>&1 | tail -18
