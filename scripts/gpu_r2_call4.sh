mkdir -p gpurun_out/r02
rm -f gpurun_out/parity_allowance.jsonl
timeout 900 python -m pytest tests -m gpu -q --timeout=300 -p no:cacheprovider > gpurun_out/r02/pytest_gpu_call4.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r02/pytest_gpu_call4.log
grep -E "^(FAILED|ERROR)|passed|failed|rc=" gpurun_out/r02/pytest_gpu_call4.log | cut -c1-300 | head -40
grep -E "^E  +" gpurun_out/r02/pytest_gpu_call4.log | cut -c1-300 | head -40
{
for Q in 1 2 3 4; do
  QUEUES=$Q python scripts/bench_world.py football 131072 300
  QUEUES=$Q python scripts/bench_world.py balance 1048576 100
  QUEUES=$Q python scripts/bench_world.py balance 32768 3000
  QUEUES=$Q python scripts/bench_world.py navigation 65536 1000
done
} 2>&1 | grep -v amdgpu.ids > gpurun_out/r02/ab_call4.log
cat gpurun_out/r02/ab_call4.log
