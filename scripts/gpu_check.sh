# one GPU round: parity tests, smoke, bench (+ lanes sweep)
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/smoke.log
timeout 1800 python -m pytest tests -m gpu -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
timeout 300 python bench.py > gpurun_out/bench.log 2>&1; echo "bench rc=$?" >> gpurun_out/bench.log
rm -f gpurun_out/bench_lanes.log
for L in 2 4 8 12 16; do timeout 120 python bench.py --lanes $L --no-cpu-baseline --steps 1000 --warmup 100 2>/dev/null | grep '^{' >> gpurun_out/bench_lanes.log; done
tail -2 gpurun_out/smoke.log
grep -E "^(FAILED|ERROR)|passed|failed|rc=" gpurun_out/pytest_gpu.log | cut -c1-300
grep -E "^E  +(Assertion|.*Error)" gpurun_out/pytest_gpu.log | cut -c1-330 | head -30
grep '^{' gpurun_out/bench.log | cut -c1-1600
python - <<'PY'
import json
for l in open('gpurun_out/bench_lanes.log'):
    d=json.loads(l); print('lanes',d['config']['lanes_per_env'], 'env-steps/s %.3e'%d['value'], 'kernel_us %.2f'%d['roofline']['kernel_us'], 'hbm frac %.3f'%d['roofline']['frac'])
PY
