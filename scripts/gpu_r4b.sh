# Round 4, second GPU call: tests on the new build, football contact-list size A/B (3 resident tiles per CU), navigation
# observation writer split, attached-reference leg, counters of the latency-regime shards
TAG=r04
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/${TAG}b; mkdir -p $OUT
cd $R
S=$R/scripts
timeout 1500 python -m pytest tests -m gpu -q --timeout=600 -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log
grep -E "^(FAILED|ERROR)|passed|failed|rc=|needed it|fixtures with any" $OUT/pytest_gpu.log | cut -c1-300 | head -20
grep -E "^E  +(Assertion|.*Error)" $OUT/pytest_gpu.log | cut -c1-300 | head -20
{
for LIB in libvmas_hip.so libvmas_hip_cap128.so; do
  export VMAS_HIP_LIB=$LIB
  for Q in 1 2; do FORCES=random QUEUES=$Q python $S/bench_world.py football 131072 300; done
  FORCES=random QUEUES=1 python $S/bench_world.py football 16384 300
  FORCES=fixed QUEUES=1 python $S/bench_world.py football 131072 300
  REPS=5 python $S/bench_rollout_env.py football 131072 50
  python $S/bench_rollout_env.py football 16384 50
done
unset VMAS_HIP_LIB
} 2>&1 | grep "^{" > $OUT/${TAG}b_football_cap_ab.jsonl
cat $OUT/${TAG}b_football_cap_ab.jsonl
{ for B in 8192 16384 65536; do ACTIONS=zero python $S/bench_bound.py navigation $B; done; python $S/bench_rollout_env.py navigation 8192 50; python $S/bench_bound.py balance 32768; python $S/bench_bound.py transport 16384; } 2>&1 | grep "^{" > $OUT/${TAG}b_env_step_bound_rates.jsonl
cat $OUT/${TAG}b_env_step_bound_rates.jsonl
python bench.py --no-cpu-baseline --no-other-configs --steps 500 --warmup 50 2>$OUT/bench_attached.err | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'): d=json.loads(l); print(json.dumps({'config':'balance','attached_reference':d.get('attached_reference')}))" > $OUT/${TAG}b_attached_reference.jsonl
python bench.py --config navigation --no-cpu-baseline --steps 300 --warmup 50 2>>$OUT/bench_attached.err | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'): d=json.loads(l); print(json.dumps({'config':'navigation','attached_reference':d.get('attached_reference')}))" >> $OUT/${TAG}b_attached_reference.jsonl
cat $OUT/${TAG}b_attached_reference.jsonl; tail -3 $OUT/bench_attached.err
export EVIDENCE_DIR=${TAG}b
ACTIONS=zero RATED=step_kernel_spec_multi:env bash scripts/gpu_counters.sh ${TAG}b_navigation8192_env_step 1480 30000 8192 -- python $S/bench_bound.py navigation 8192 > /dev/null 2>&1
FORCES=random RATED=step_kernel_compact:physics bash scripts/gpu_counters.sh ${TAG}b_football16384_physics_compact 948 11900 16384 -- python $S/bench_world.py football 16384 300 > /dev/null 2>&1
RATED=step_kernel_spec_multi:env bash scripts/gpu_counters.sh ${TAG}b_balance32768_env_step 657 2000 32768 -- python $S/bench_bound.py balance 32768 > /dev/null 2>&1
grep -h "== \|sustained\|traffic / alg\|share of wave\|duration" $OUT/${TAG}b_navigation8192_env_step_pmc_summary.txt $OUT/${TAG}b_football16384_physics_compact_pmc_summary.txt $OUT/${TAG}b_balance32768_env_step_pmc_summary.txt | cut -c1-220
