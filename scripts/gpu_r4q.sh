export COMPACT_STATS=1 QUEUES=1
for B in 131072 16384; do
for F in random fixed; do
  for ST in 100 300 600; do
    echo "== $B $F steps=$ST"; FORCES=$F python scripts/bench_world.py football $B $ST 2>&1 | grep "compact stats\|world_step_us" | cut -c1-250
  done
done
done
