cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r05i; mkdir -p $OUT
for B in 1048576 262144 32768; do for PAD in 0 1 3 0 1; do
  VMAS_LD_FROM=1024 VMAS_LD_PAD=$PAD FORCES=random QUEUES=1 python scripts/bench_world.py balance $B 300 2>/dev/null | grep "^{" | sed "s/^{/{\"ld_pad_tiles\": $PAD, /" | tee -a $OUT/r05i_ld_pad_balance.jsonl | cut -c1-60,200-300
done; done
for PAD in 0 1; do
  VMAS_LD_FROM=1024 VMAS_LD_PAD=$PAD FORCES=random python scripts/bench_world.py balance 1048576 300 2>/dev/null | grep "^{" | sed "s/^{/{\"ld_pad_tiles\": $PAD, /" | tee -a $OUT/r05i_ld_pad_balance.jsonl | cut -c1-60,200-300
  VMAS_LD_FROM=1024 VMAS_LD_PAD=$PAD COMPACT=1 FORCES=random QUEUES=1 python scripts/bench_world.py football 131072 300 2>/dev/null | grep "^{" | sed "s/^{/{\"ld_pad_tiles\": $PAD, /" | tee -a $OUT/r05i_ld_pad_balance.jsonl | cut -c1-60,200-300
  VMAS_LD_FROM=1024 VMAS_LD_PAD=$PAD FORCES=random QUEUES=1 python scripts/bench_world.py navigation 65536 300 2>/dev/null | grep "^{" | sed "s/^{/{\"ld_pad_tiles\": $PAD, /" | tee -a $OUT/r05i_ld_pad_balance.jsonl | cut -c1-60,200-300
done
