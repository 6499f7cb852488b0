# A/B: VMAS_SHARE = 0 (every side evaluates its own copy) | 1 (pairs of two dynamic entities once, except sphere-sphere) | 2 (all)
# | unset: the library's occupancy rule
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q -x > gpurun_out/pytest_ab.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_ab.log
grep -E "^(FAILED|ERROR)|passed|failed|rc=" gpurun_out/pytest_ab.log | cut -c1-300
grep -E "^E  +(Assertion|.*Error)" gpurun_out/pytest_ab.log | cut -c1-330 | head -20
for M in 0 auto; do
  [ $M = auto ] && unset VMAS_SHARE || export VMAS_SHARE=$M
  python bench.py --no-cpu-baseline --no-fused --steps 3000 --warmup 300 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('balance 32768 share=$M kernel_us %.2f'%d['roofline']['kernel_us'])"
done
for s in "transport 16384" "navigation 65536" "navigation 8192" "football 131072" "football 16384" "balance 1048576"; do for M in 0 auto; do
  [ $M = auto ] && unset VMAS_SHARE || export VMAS_SHARE=$M
  echo "share=$M $(python scripts/bench_world.py $s 2>/dev/null | tail -1)"
done; done
unset VMAS_SHARE
for s in "balance 32768" "transport 16384" "navigation 65536" "football 131072"; do ONLY=fused-eager python scripts/bench_env.py $s | grep scenario; done
