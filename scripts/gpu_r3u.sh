export NAV_TILES=64
VMAS_TRACE=2 python scripts/trace_nav.py 65536 2>&1 | tail -16
VMAS_TRACE=2 python scripts/trace_nav.py 8192 2>&1 | tail -16
