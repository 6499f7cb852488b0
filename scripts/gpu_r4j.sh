export ACTIONS=zero
echo "== fresh"; VMAS_TRACE=2 python scripts/trace_nav.py 16384 2>&1 | grep "mean" | cut -c1-90
echo "== later"; VMAS_TRACE=2 python scripts/trace_nav.py 16384 2>&1 | grep "mean" | cut -c1-90
