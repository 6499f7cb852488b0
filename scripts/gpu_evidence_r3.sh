# Round 3's evidence in one GPU call (summaries land in gpurun_out/r03/, copy what is to be judged into profiles/):
#   bash scripts/gpu_evidence_r3.sh [skip-tests]
TAG=r03
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd $R
if [ "${1:-}" != "skip-tests" ]; then
rm -f gpurun_out/parity_allowance.jsonl
timeout 1500 python -m pytest tests -m gpu -q --timeout=600 -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log
grep -E "^(FAILED|ERROR)|passed|failed|rc=" $OUT/pytest_gpu.log | cut -c1-300 | head -20
cp gpurun_out/parity_allowance.jsonl $OUT/${TAG}_parity_allowance.jsonl 2>/dev/null
cp gpurun_out/broad_phase_full_size.jsonl $OUT/${TAG}_broad_phase_full_size.jsonl 2>/dev/null
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
fi
# bench lines: default (reference CPU baseline in the same run) and the driver's invocation
python bench.py > $OUT/${TAG}_bench_line_default.json 2> $OUT/bench_default.err; tail -c 300 $OUT/${TAG}_bench_line_default.json; echo
python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $OUT/${TAG}_bench_line_driver_style.json 2> $OUT/bench_driver_style.err
# rocprofv3 of the bench command (one queue: clean per-kernel durations + PMC + traffic; two queues: kernel trace)
BENCH_ARGS="--queues 1" bash scripts/gpu_prof.sh > $OUT/prof_q1.log 2>&1
f=$(find gpurun_out/prof/trace -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && grep -v "at::\|rocclr\|Cijk" "$f" | head -8 > $OUT/${TAG}_bench_q1_kernel_stats.csv
cp gpurun_out/prof/latest_traffic.json $OUT/latest_traffic.json 2>/dev/null
grep "per-dispatch mean" $OUT/prof_q1.log > $OUT/${TAG}_bench_q1_pmc_raw.txt
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/q2; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/q2 -o bench -- python $R/bench.py --no-cpu-baseline --no-fused --steps 1000 --warmup 100 > /tmp/q2.log 2>&1
f=$(find /tmp/q2 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && grep -v "at::\|rocclr\|Cijk" "$f" | head -8 > $OUT/${TAG}_bench_q2_kernel_stats.csv
cd $R
S=$R/scripts
export QUEUES=1 EVIDENCE_DIR=$TAG
# counters: the lane-compacted football kernel (random-action protocol) and the interpreter beside it; balance 1 M;
# the one-launch navigation step; the one-launch balance step
FORCES=random RATED=step_kernel_compact bash scripts/gpu_counters.sh ${TAG}_football131072_physics_compact 948 11900 131072 -- python $S/bench_world.py football 131072 200 > /dev/null 2>&1
FORCES=random COMPACT=0 bash scripts/gpu_counters.sh ${TAG}_football131072_physics_interpreter 948 11900 131072 -- python $S/bench_world.py football 131072 200 > /dev/null 2>&1
FORCES=random RATED=step_kernel_compact bash scripts/gpu_counters.sh ${TAG}_football16384_physics_compact 948 11900 16384 -- python $S/bench_world.py football 16384 300 > /dev/null 2>&1
bash scripts/gpu_counters.sh ${TAG}_balance1048576_physics 384 1700 1048576 -- python $S/bench_world.py balance 1048576 100 > /dev/null 2>&1
bash scripts/gpu_counters.sh ${TAG}_balance32768_physics 384 1700 32768 -- python $S/bench_world.py balance 32768 300 > /dev/null 2>&1
ACTIONS=zero RATED=step_kernel_spec_multi bash scripts/gpu_counters.sh ${TAG}_navigation65536_env_step 1480 30000 65536 -- python $S/bench_bound.py navigation 65536 > /dev/null 2>&1
ACTIONS=zero RATED=step_kernel_spec_multi bash scripts/gpu_counters.sh ${TAG}_navigation8192_env_step 1480 30000 8192 -- python $S/bench_bound.py navigation 8192 > /dev/null 2>&1
RATED=step_kernel_spec_multi bash scripts/gpu_counters.sh ${TAG}_balance32768_env_step 657 2000 32768 -- python $S/bench_bound.py balance 32768 > /dev/null 2>&1
unset QUEUES
# rates
{
for W in "balance 32768" "balance 65536" "balance 131072" "balance 1048576" "transport 16384" "navigation 65536" "navigation 8192"; do
  for Q in 1 2; do QUEUES=$Q python scripts/bench_world.py $W 500; done
done
for W in "football 131072" "football 16384"; do
  for Q in 1 2; do for CP in 1 0; do FORCES=random COMPACT=$CP QUEUES=$Q python scripts/bench_world.py $W 300; done; done
  for CP in 1 0; do FORCES=fixed COMPACT=$CP QUEUES=1 python scripts/bench_world.py $W 300; done
done
} 2>&1 | grep "^{" > $OUT/${TAG}_world_step_rates.jsonl
{ for W in "balance 32768" "transport 16384" "navigation 65536" "navigation 8192" "football 131072" "football 16384"; do ONLY=fused-eager python scripts/bench_env.py $W; ONLY=fused-graph python scripts/bench_env.py $W; done; } 2>&1 | grep "^{" > $OUT/${TAG}_env_step_rates.jsonl
{ for B in 8192 16384 32768 65536 131072; do ACTIONS=zero python scripts/bench_bound.py navigation $B; done; python scripts/bench_bound.py balance 32768; python scripts/bench_bound.py balance 65536; python scripts/bench_bound.py transport 16384; } 2>&1 | grep "^{" > $OUT/${TAG}_env_step_bound_rates.jsonl
{ python scripts/bench_rollout_env.py balance 32768 100; python scripts/bench_rollout_env.py transport 16384 100; python scripts/bench_rollout_env.py navigation 8192 50; REPS=5 python scripts/bench_rollout_env.py football 131072 50; python scripts/bench_rollout_env.py football 16384 50; } 2>&1 | grep "^{" > $OUT/${TAG}_env_rollout_rates.jsonl
python scripts/bench_specialize.py 2>&1 | grep "^{" > $OUT/${TAG}_runtime_specialisation_rates.jsonl
VMAS_TRACE=2 NAV_TILES=64 python scripts/trace_nav.py 65536 > $OUT/${TAG}_navigation65536_env_step_phase_trace.txt 2>&1
VMAS_TRACE=2 NAV_TILES=64 python scripts/trace_nav.py 8192 > $OUT/${TAG}_navigation8192_env_step_phase_trace.txt 2>&1
tail -3 $OUT/${TAG}_world_step_rates.jsonl; cat $OUT/${TAG}_env_rollout_rates.jsonl; cat $OUT/${TAG}_env_step_bound_rates.jsonl
