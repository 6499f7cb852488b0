set -x
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/smoke.log
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
timeout 300 python bench.py > gpurun_out/bench.log 2>&1; echo "bench rc=$?" >> gpurun_out/bench.log
for L in 1 2 4 8 16 32 64; do timeout 120 python bench.py --lanes $L --no-cpu-baseline --steps 1000 --warmup 100 >> gpurun_out/bench_lanes.log 2>&1; done
tail -5 gpurun_out/smoke.log; tail -15 gpurun_out/pytest_gpu.log; cat gpurun_out/bench.log | cut -c1-1500; cat gpurun_out/bench_lanes.log | python -c "
import sys, json
for l in sys.stdin:
    try:
        d=json.loads(l); print(d['config']['lanes_per_env'], d['value'], d['roofline']['kernel_us'], d['roofline']['frac'])
    except Exception as e: print('ERR', l[:200])
"
