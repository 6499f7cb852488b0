# Round 4, fifth GPU call: tests (the exact broad phase's grid barrier is now one atomic per tile and word, the mask read
# from LDS); navigation with the barrier collected behind the observation writer; the exact form's cost
TAG=r04e
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd $R
S=$R/scripts
timeout 1500 python -m pytest tests -m gpu -q --timeout=600 -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log
grep -E "^(FAILED|ERROR)|passed|failed|rc=|needed it|fixtures with any" $OUT/pytest_gpu.log | cut -c1-300 | head -20
grep -E "^E  +(Assertion|.*Error)" $OUT/pytest_gpu.log | cut -c1-300 | head -20
{ for B in 2048 8192 16384 65536; do ACTIONS=zero python $S/bench_bound.py navigation $B; done; python $S/bench_bound.py navigation 8192; python $S/bench_rollout_env.py navigation 8192 50; } 2>&1 | grep "^{" > $OUT/${TAG}_navigation_rates.jsonl
cat $OUT/${TAG}_navigation_rates.jsonl
python $S/bench_exact.py 300 2>&1 | grep "^{" > $OUT/${TAG}_exact_in_launch_cost.jsonl; cat $OUT/${TAG}_exact_in_launch_cost.jsonl
VMAS_TRACE=2 python scripts/trace_nav.py 8192 2>&1 | grep -v amdgpu > $OUT/${TAG}_navigation8192_env_step_phase_trace.txt; cat $OUT/${TAG}_navigation8192_env_step_phase_trace.txt
