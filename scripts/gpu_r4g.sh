# Round 4, seventh GPU call: (i) HBM's rate for football's observation tensor by store pattern (scripts/micro/store_pattern.hip);
# (ii) tests + rates after three load-phase changes: the compacted kernel's trig-cache entries as vector loads (were four
# serialized scalar round trips), the lean headline kernel's first agent-force rows requested in front of the entity loop,
# the env-step kernel's torque row in front of the action branch
TAG=r04g
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd $R
S=$R/scripts
{ timeout 120 $S/micro/store_pattern 131072 50; timeout 60 $S/micro/store_pattern 16384 200; } > $OUT/${TAG}_store_pattern.jsonl 2>&1; cat $OUT/${TAG}_store_pattern.jsonl
timeout 1500 python -m pytest tests -m gpu -q --timeout=600 -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log
grep -E "^(FAILED|ERROR)|passed|failed|rc=|needed it|fixtures with any" $OUT/pytest_gpu.log | cut -c1-300 | head -20
grep -E "^E  +(Assertion|.*Error)" $OUT/pytest_gpu.log | cut -c1-300 | head -20
python bench.py --no-cpu-baseline --no-other-configs --no-attached > $OUT/${TAG}_bench_line_balance.json 2> $OUT/bench.err; cut -c1-1500 $OUT/${TAG}_bench_line_balance.json
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs --no-attached 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print({k: d[k] for k in ('value','ms_per_step','steps')}, d['roofline'])"
{ for Q in 1 2; do COMPACT=1 FORCES=random QUEUES=$Q python $S/bench_world.py football 131072 300; done; COMPACT=1 FORCES=random python $S/bench_world.py football 16384 300; FORCES=random python $S/bench_world.py football 16384 300; } 2>&1 | grep "^{" > $OUT/${TAG}_football_rates.jsonl; cat $OUT/${TAG}_football_rates.jsonl
{ REPS=5 python $S/bench_rollout_env.py football 131072 50; REPS=5 python $S/bench_rollout_env.py football 16384 50; python $S/bench_bound.py football 131072; python $S/bench_bound.py balance 32768; python $S/bench_rollout_env.py balance 32768 100; python $S/bench_bound.py transport 16384; } 2>&1 | grep "^{" > $OUT/${TAG}_env_rates.jsonl; cat $OUT/${TAG}_env_rates.jsonl
python $S/trace_compact.py football 16384 2>&1 | grep -v amdgpu > $OUT/${TAG}_football16384_compact_phase_trace.txt; tail -25 $OUT/${TAG}_football16384_compact_phase_trace.txt
