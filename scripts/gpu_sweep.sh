for B in 8192 32768 131072 524288 2097152; do for L in 2 4 8; do
python bench.py --no-cpu-baseline --no-fused --steps 300 --warmup 50 --num-envs $B --lanes $L 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('B $B lanes',d['config']['lanes_per_env'],'kernel_us %.2f'%d['roofline']['kernel_us'],'env-steps/s %.3e'%d['value'],'hbm frac %.3f'%d['roofline']['frac'])"
done; done
