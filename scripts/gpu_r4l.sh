python scripts/window_rate_bound.py 16384 25
python scripts/window_rate_bound.py 16384 10
