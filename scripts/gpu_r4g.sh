export ACTIONS=zero
c() { rocm-smi --showclocks 2>/dev/null | grep -i "sclk\|mclk" | head -2 | tr '\n' ' '; echo; }
c
for i in 1 2 3; do python scripts/bench_bound.py navigation 16384 | tail -1 | cut -c60-200; c; done
sleep 15; c
python scripts/bench_bound.py navigation 16384 | tail -1 | cut -c60-200
python scripts/bench_bound.py balance 32768 | tail -1 | cut -c60-200
python scripts/bench_bound.py navigation 16384 | tail -1 | cut -c60-200
