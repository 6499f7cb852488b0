# waves per tile ("lanes per env") sweep with the current kernel
for L in 4 6 8 12 16; do python bench.py --lanes $L --no-cpu-baseline --no-fused --steps 2000 --warmup 200 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('balance 32768 lanes $L kernel_us %.2f'%d['roofline']['kernel_us'])"; done
for s in "transport 16384" "navigation 65536" "navigation 8192" "football 131072" "football 16384" "balance 1048576" "balance 131072"; do for L in 2 4 8; do
  echo "$(LANES=$L python scripts/bench_world.py $s 2>/dev/null | tail -1 | cut -c1-110)"
done; done
