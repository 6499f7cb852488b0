# The round's evidence in one GPU call (summaries land in gpurun_out/$TAG/, copy what is to be judged into profiles/):
#   bash scripts/gpu_evidence.sh r02
TAG=${1:-r02}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd $R
# 1. the GPU test-suite
rm -f gpurun_out/parity_allowance.jsonl
timeout 900 python -m pytest tests -m gpu -q --timeout=300 -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log
grep -E "^(FAILED|ERROR)|passed|failed|rc=" $OUT/pytest_gpu.log | cut -c1-300 | head -20
cp gpurun_out/parity_allowance.jsonl $OUT/parity_allowance.jsonl 2>/dev/null
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
# 2. the bench lines: default (reference CPU baseline in the same run), the driver's invocation, one queue
python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; tail -c 400 $OUT/bench_default.json; echo
python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_driver_style.json 2> $OUT/bench_driver_style.err
# 3. rocprofv3 of the bench command: kernel trace + PMC passes + HBM traffic (one queue: clean per-kernel durations; then
#    the default two queues, kernel trace only)
BENCH_ARGS="--queues 1" bash scripts/gpu_prof.sh > $OUT/prof_q1.log 2>&1
f=$(find gpurun_out/prof/trace -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && grep -v "at::\|rocclr\|Cijk" "$f" | head -8 > $OUT/${TAG}_bench_q1_kernel_stats.csv
cp gpurun_out/prof/latest_traffic.json $OUT/latest_traffic.json 2>/dev/null
grep "per-dispatch mean" $OUT/prof_q1.log > $OUT/${TAG}_bench_q1_pmc_raw.txt
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/q2; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/q2 -o bench -- python $R/bench.py --no-cpu-baseline --no-fused --steps 1000 --warmup 100 > /tmp/q2.log 2>&1
f=$(find /tmp/q2 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && grep -v "at::\|rocclr\|Cijk" "$f" | head -8 > $OUT/${TAG}_bench_q2_kernel_stats.csv
cd $R
# 4. counters of the other kernels (physics of football / navigation / balance 1 M, LIDAR)
S=$R/scripts
export QUEUES=1 EVIDENCE_DIR=$TAG
bash scripts/gpu_counters.sh ${TAG}_football131072_physics 948 11900 131072 -- python $S/bench_world.py football 131072 200 > /dev/null 2>&1
bash scripts/gpu_counters.sh ${TAG}_navigation65536_physics 672 1800 65536 -- python $S/bench_world.py navigation 65536 200 > /dev/null 2>&1
bash scripts/gpu_counters.sh ${TAG}_transport16384_physics 312 800 16384 -- python $S/bench_world.py transport 16384 300 > /dev/null 2>&1
bash scripts/gpu_counters.sh ${TAG}_balance1048576_physics 384 1700 1048576 -- python $S/bench_world.py balance 1048576 100 > /dev/null 2>&1
bash scripts/gpu_counters.sh ${TAG}_balance32768_physics 384 1700 32768 -- python $S/bench_world.py balance 32768 300 > /dev/null 2>&1
SPEC=0 bash scripts/gpu_counters.sh ${TAG}_balance32768_physics_interpreter 384 1700 32768 -- python $S/bench_world.py balance 32768 300 > /dev/null 2>&1
SPEC=0 bash scripts/gpu_counters.sh ${TAG}_balance1048576_physics_interpreter 384 1700 1048576 -- python $S/bench_world.py balance 1048576 100 > /dev/null 2>&1
bash scripts/gpu_counters.sh ${TAG}_lidar65536 480 27000 65536 -- python $S/bench_lidar.py 65536 > /dev/null 2>&1
unset QUEUES
# 5. rates: World.step by queues, Environment.step eager / graph, Environment.rollout
{
for W in "balance 32768" "transport 16384" "navigation 65536" "navigation 8192" "football 131072" "football 16384" "balance 1048576"; do
  for Q in 1 2; do QUEUES=$Q python scripts/bench_world.py $W 500; done
done
for W in "balance 32768" "balance 131072" "balance 1048576" "transport 16384" "navigation 65536" "navigation 8192"; do   # specialised kernel vs interpreter, one queue
  for SP in 1 0; do SPEC=$SP QUEUES=1 python scripts/bench_world.py $W 1000; done
done
} 2>&1 | grep "^{" > $OUT/${TAG}_world_step_rates.jsonl
{ ONLY=fused-eager python scripts/bench_env.py balance 32768; ONLY=fused-graph python scripts/bench_env.py balance 32768; ONLY=fused-eager python scripts/bench_env.py transport 16384; ONLY=fused-eager python scripts/bench_env.py navigation 65536; ONLY=fused-eager python scripts/bench_env.py football 131072; } 2>&1 | grep "^{" > $OUT/${TAG}_env_step_rates.jsonl
{ python scripts/bench_rollout_env.py balance 32768 100; python scripts/bench_rollout_env.py transport 16384 100; } 2>&1 | grep "^{" > $OUT/${TAG}_env_rollout_rates.jsonl
# 6. instruction counts by phase (profile build)
bash scripts/gpu_ablate.sh balance 32768 300 > $OUT/${TAG}_balance32768_by_phase.txt 2>&1
bash scripts/gpu_ablate.sh football 131072 100 > $OUT/${TAG}_football131072_by_phase.txt 2>&1
tail -3 $OUT/${TAG}_world_step_rates.jsonl; cat $OUT/${TAG}_env_rollout_rates.jsonl; cat $OUT/${TAG}_balance32768_by_phase.txt
