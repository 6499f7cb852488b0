export ACTIONS=zero DIAG=1
python scripts/bench_bound.py navigation 16384 2>&1 | grep "DIAG\|step_bound" | cut -c1-200
python scripts/bench_bound.py navigation 16384 2>&1 | grep "DIAG\|step_bound" | cut -c1-200
