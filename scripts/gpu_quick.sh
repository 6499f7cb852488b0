# parity tests + headline kernel time + physics-only rates of the other benchmark worlds
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q -x > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
grep -E "^(FAILED|ERROR)|passed|failed|rc=" gpurun_out/pytest_gpu.log | cut -c1-300
grep -E "^E  +(Assertion|.*Error)" gpurun_out/pytest_gpu.log | cut -c1-330 | head -20
for rep in 1 2; do
python bench.py --no-cpu-baseline --no-fused --steps 3000 --warmup 300 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('balance 32768 kernel_us %.2f'%d['roofline']['kernel_us'])"
done
for s in "transport 16384" "navigation 65536" "navigation 8192" "football 131072" "football 16384" "balance 1048576"; do
  python scripts/bench_world.py $s 2>/dev/null | tail -1
done
[ -n "$ENVSTEP" ] && for s in "balance 32768" "transport 16384" "navigation 65536" "football 131072"; do ONLY=fused-eager python scripts/bench_env.py $s | grep scenario; done
[ -n "$LIDAR" ] && python scripts/bench_lidar.py 2>/dev/null | tail -3
true
