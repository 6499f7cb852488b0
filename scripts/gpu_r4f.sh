export ACTIONS=zero
python scripts/bench_bound.py navigation 16384 | tail -1
python scripts/host_rate_bound.py | tail -6
python scripts/bench_bound.py navigation 16384 | tail -1
python scripts/bench_bound.py navigation 8192 | tail -1
timeout 600 python -m pytest tests/test_specialize_gpu.py -q --timeout=600 -p no:cacheprovider -m gpu 2>&1 | tail -2
