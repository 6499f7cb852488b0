export COMPACT_STATS=1 QUEUES=1 FORCES=fixed
for ST in 20 40 70 100 150; do
  for CP in 1 0; do
    echo -n "steps=$ST COMPACT=$CP "; COMPACT=$CP python scripts/bench_world.py football 131072 $ST 2>&1 | grep "compact stats\|world_step_us" | sed 's/.*overflowed/contacts/; s/.*world_step_us/us/' | tr '\n' ' '; echo
  done
done
