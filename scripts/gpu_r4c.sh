# Round 4, third GPU call: tests; football contact-list capacity A/B with the compacted kernel PINNED (the second call's run
# was on the interpreter for most of its launches: the adaptive choice had backed off during the warm-up); navigation with
# the pair bits published by the finding wave; the bench line with the gate in front of its timed windows
TAG=r04c
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd $R
S=$R/scripts
timeout 1500 python -m pytest tests -m gpu -q --timeout=600 -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log
grep -E "^(FAILED|ERROR)|passed|failed|rc=|needed it|fixtures with any" $OUT/pytest_gpu.log | cut -c1-300 | head -20
grep -E "^E  +(Assertion|.*Error)" $OUT/pytest_gpu.log | cut -c1-300 | head -20
{
for LIB in libvmas_hip.so libvmas_hip_cap128.so; do
  export VMAS_HIP_LIB=$LIB
  for Q in 1 2; do COMPACT=1 FORCES=random QUEUES=$Q python $S/bench_world.py football 131072 300; done
  COMPACT=1 FORCES=random QUEUES=1 python $S/bench_world.py football 16384 300
  COMPACT=1 FORCES=random QUEUES=1 python $S/bench_world.py football 32768 300
  REPS=5 python $S/bench_rollout_env.py football 131072 50
done
unset VMAS_HIP_LIB
COMPACT=0 FORCES=random QUEUES=1 python $S/bench_world.py football 131072 300
COMPACT=0 FORCES=random QUEUES=1 python $S/bench_world.py football 16384 300
} 2>&1 | grep "^{" > $OUT/${TAG}_football_cap_ab.jsonl
cat $OUT/${TAG}_football_cap_ab.jsonl
{ for B in 8192 16384 65536; do ACTIONS=zero python $S/bench_bound.py navigation $B; done; python $S/bench_rollout_env.py navigation 8192 50; python $S/bench_bound.py balance 32768; python $S/bench_bound.py transport 16384; } 2>&1 | grep "^{" > $OUT/${TAG}_env_step_bound_rates.jsonl
cat $OUT/${TAG}_env_step_bound_rates.jsonl
python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs --no-attached > $OUT/${TAG}_bench_line_driver_style.json 2>$OUT/bench.err
python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs --no-attached --gate-us 0 > $OUT/${TAG}_bench_line_driver_style_no_gate.json 2>>$OUT/bench.err
python bench.py --no-cpu-baseline --no-other-configs --no-attached > $OUT/${TAG}_bench_line_default_nocpu.json 2>>$OUT/bench.err
for f in driver_style driver_style_no_gate default_nocpu; do python -c "
import json; d=json.loads(open('$OUT/${TAG}_bench_line_$f.json').read().strip().splitlines()[-1]); e=d['environment_step']
print('$f', 'ms_per_step', d['ms_per_step'], 'frac', d['roofline']['frac'], 'env us', e['us_per_step'], e['gpu_us_per_step'], 'wall', d['wall']['ms_per_step'])"; done
VMAS_TRACE=2 python scripts/trace_nav.py 8192 2>&1 | grep -v amdgpu > $OUT/${TAG}_navigation8192_env_step_phase_trace.txt; cat $OUT/${TAG}_navigation8192_env_step_phase_trace.txt
