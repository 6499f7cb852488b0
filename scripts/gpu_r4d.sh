# Round 4, fourth GPU call: tests; the navigation grid barrier as ONE atomic per tile; football with 256 contact slots per
# tile (three resident tiles per CU) as the default
TAG=r04d
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd $R
S=$R/scripts
timeout 1500 python -m pytest tests -m gpu -q --timeout=600 -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log
grep -E "^(FAILED|ERROR)|passed|failed|rc=|needed it|fixtures with any" $OUT/pytest_gpu.log | cut -c1-300 | head -20
grep -E "^E  +(Assertion|.*Error)" $OUT/pytest_gpu.log | cut -c1-300 | head -20
{ for B in 2048 8192 16384 65536; do ACTIONS=zero python $S/bench_bound.py navigation $B; done; python $S/bench_bound.py navigation 8192; python $S/bench_rollout_env.py navigation 8192 50; } 2>&1 | grep "^{" > $OUT/${TAG}_navigation_rates.jsonl
cat $OUT/${TAG}_navigation_rates.jsonl
{
for Q in 1 2; do COMPACT=1 FORCES=random QUEUES=$Q python $S/bench_world.py football 131072 300; done
FORCES=random QUEUES=1 python $S/bench_world.py football 131072 300
COMPACT=1 FORCES=fixed QUEUES=1 python $S/bench_world.py football 131072 300
COMPACT=1 FORCES=random QUEUES=1 python $S/bench_world.py football 16384 300
REPS=5 python $S/bench_rollout_env.py football 131072 50
python $S/bench_rollout_env.py football 16384 50
} 2>&1 | grep "^{" > $OUT/${TAG}_football_rates.jsonl
cat $OUT/${TAG}_football_rates.jsonl
python bench.py --config football --no-cpu-baseline --no-attached --steps 1000 --warmup 100 > $OUT/${TAG}_bench_line_football_nocpu.json 2>$OUT/bench.err
python bench.py --config navigation --no-cpu-baseline --no-attached --steps 1000 --warmup 100 > $OUT/${TAG}_bench_line_navigation_nocpu.json 2>>$OUT/bench.err
for f in football navigation; do python -c "
import json; d=json.loads(open('$OUT/${TAG}_bench_line_${f}_nocpu.json').read().strip().splitlines()[-1]); e=d['environment_step']
print('$f', 'us_per_step', d['ms_per_step']*1e3, 'frac', d['roofline']['frac'], 'queues', d['config']['queues'], 'env us', e.get('us_per_step'), e.get('gpu_us_per_step'), 'bound', e.get('bound',{}).get('gpu_us_per_step'), 'rollout', e.get('rollout',{}).get('us_per_step'), 'single', d.get('single_queue',{}).get('us_per_step'))"; done
VMAS_TRACE=2 python scripts/trace_nav.py 8192 2>&1 | grep -v amdgpu > $OUT/${TAG}_navigation8192_env_step_phase_trace.txt; cat $OUT/${TAG}_navigation8192_env_step_phase_trace.txt
