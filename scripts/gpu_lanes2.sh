for s in "transport 16384" "navigation 8192" "football 131072" "football 16384" "navigation 65536" "balance 16384"; do for L in 8 16; do
  echo "$(LANES=$L VMAS_DEBUG_SCHED=1 python scripts/bench_world.py $s 2>&1 | grep "LDS\|scenario" | tail -2 | cut -c1-110 | tr '\n' ' ')"
done; done
