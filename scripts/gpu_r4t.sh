timeout 900 python -m pytest tests/test_compact_gpu.py tests/test_hip_parity.py -q --timeout=600 -p no:cacheprovider -x -k "compact or queues or football" 2>&1 | grep -v "^$" | tail -3
export COMPACT_STATS=1
for F in random fixed; do
  for CP in "" 0; do
    echo -n "q=2 $F COMPACT=${CP:-auto} "; QUEUES=2 FORCES=$F COMPACT=$CP python scripts/bench_world.py football 131072 600 2>&1 | grep "compact stats\|world_step_us" | sed 's/.*world_step_us/us/' | tr '\n' ' ' | cut -c1-200; echo
  done
done
