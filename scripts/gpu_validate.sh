# HEAD validation: smoke, GPU parity tests, bench line (with timings of each leg)
mkdir -p gpurun_out
t0=$(date +%s)
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/smoke.log
t1=$(date +%s)
timeout 1500 python -m pytest tests -m gpu -q -x > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
t2=$(date +%s)
timeout 300 python bench.py > gpurun_out/bench.log 2>&1; echo "bench rc=$?" >> gpurun_out/bench.log
t3=$(date +%s)
echo "smoke $((t1-t0))s pytest $((t2-t1))s bench $((t3-t2))s"
tail -2 gpurun_out/smoke.log
grep -E "^(FAILED|ERROR)|passed|failed|rc=" gpurun_out/pytest_gpu.log | cut -c1-300
grep -E "^E  +(Assertion|.*Error)" gpurun_out/pytest_gpu.log | cut -c1-330 | head -30
grep '^{' gpurun_out/bench.log | cut -c1-2500
