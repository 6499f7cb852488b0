mkdir -p gpurun_out/r02
rm -f gpurun_out/parity_allowance.jsonl
timeout 900 python -m pytest tests -m gpu -q --timeout=300 -p no:cacheprovider > gpurun_out/r02/pytest_gpu_call3.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r02/pytest_gpu_call3.log
grep -E "^(FAILED|ERROR)|passed|failed|rc=" gpurun_out/r02/pytest_gpu_call3.log | cut -c1-300 | head -40
grep -E "^E  +" gpurun_out/r02/pytest_gpu_call3.log | cut -c1-300 | head -40
{
python scripts/bench_rollout_env.py balance 32768 100
python scripts/bench_rollout_env.py transport 16384 100
for Q in 1 2; do
  QUEUES=$Q python scripts/bench_world.py football 131072 300
  QUEUES=$Q python scripts/bench_world.py balance 1048576 100
  QUEUES=$Q python scripts/bench_world.py balance 131072 500
  QUEUES=$Q python scripts/bench_world.py balance 32768 3000
done
} 2>&1 | grep -v amdgpu.ids > gpurun_out/r02/ab_call3.log
cat gpurun_out/r02/ab_call3.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
