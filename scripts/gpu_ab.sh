# A/B of kernel variants inside one session: VMAS_ABLATE bits, 3 repeats each
for rep in 1 2 3; do for A in ${ABL:-0 64}; do
  VMAS_ABLATE=$A python bench.py --no-cpu-baseline --steps 2000 --warmup 200 ${BENCH_ARGS:-} 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ablate $A  kernel_us %.2f'%d['roofline']['kernel_us'])"
done; done
