timeout 900 python -m pytest tests/test_compact_gpu.py -q --timeout=600 -p no:cacheprovider -m gpu -x -s 2>&1 | grep -v "^$" | tail -5
p() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['num_envs'], d['forces'], 'compact', d['compact'], 'queues', d['queues'], 'us', d['world_step_us'])"; }
for F in fixed random; do
  for CP in "" 1 0; do
    echo -n "COMPACT=${CP:-auto} "; FORCES=$F COMPACT=$CP QUEUES=1 python scripts/bench_world.py football 131072 600 | tail -1 | p football
  done
done
for CP in "" 1 0; do echo -n "COMPACT=${CP:-auto} "; FORCES=fixed COMPACT=$CP QUEUES=1 python scripts/bench_world.py football 16384 600 | tail -1 | p football; done
