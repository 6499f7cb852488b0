timeout 900 python -m pytest tests/test_compact_gpu.py tests/test_abi_symbols.py -q --timeout=600 -p no:cacheprovider -x -s 2>&1 | grep -v "^$" | tail -4
export COMPACT_STATS=1
for Q in 1 2; do for F in random fixed; do
  for CP in "" 1 0; do
    echo -n "q=$Q $F COMPACT=${CP:-auto} "; QUEUES=$Q FORCES=$F COMPACT=$CP python scripts/bench_world.py football 131072 600 2>&1 | grep "compact stats\|world_step_us" | sed 's/.*world_step_us/us/' | tr '\n' ' ' | cut -c1-200; echo
  done
done; done
for CP in "" 1 0; do echo -n "16384 fixed COMPACT=${CP:-auto} "; QUEUES=1 FORCES=fixed COMPACT=$CP python scripts/bench_world.py football 16384 600 2>&1 | grep "world_step_us" | sed 's/.*world_step_us/us/'; done
