# Round 4's evidence in one GPU call (summaries land in gpurun_out/r04/, copy what is to be judged into profiles/):
#   bash scripts/gpu_evidence_r4.sh [skip-tests] [quick]
TAG=${TAG:-r04}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd $R
S=$R/scripts
if [[ " $* " != *" skip-tests "* ]]; then
rm -f gpurun_out/parity_allowance.jsonl
timeout 1500 python -m pytest tests -m gpu -q --timeout=600 -p no:cacheprovider -rs > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log
grep -E "^(FAILED|ERROR)|passed|failed|rc=" $OUT/pytest_gpu.log | cut -c1-300 | head -20
grep -E "^E  +(Assertion|.*Error)" $OUT/pytest_gpu.log | cut -c1-300 | head -20
cp gpurun_out/parity_allowance.jsonl $OUT/${TAG}_parity_allowance.jsonl 2>/dev/null
cp gpurun_out/broad_phase_full_size.jsonl $OUT/${TAG}_broad_phase_full_size.jsonl 2>/dev/null
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
fi
# bench lines: the default run (every leg: other configurations, attached reference, the reference's CPU baseline in the same
# run), the driver's invocation, and the full line of every other configuration
python bench.py > $OUT/${TAG}_bench_line_default.json 2> $OUT/bench_default.err; tail -c 400 $OUT/${TAG}_bench_line_default.json; echo; tail -3 $OUT/bench_default.err
python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $OUT/${TAG}_bench_line_driver_style.json 2> $OUT/bench_driver_style.err
if [[ " $* " != *" quick "* ]]; then
for C_ in transport transport_2pkg navigation football; do
  python bench.py --config $C_ > $OUT/${TAG}_bench_line_$C_.json 2> $OUT/bench_$C_.err; tail -c 300 $OUT/${TAG}_bench_line_$C_.json; echo; tail -2 $OUT/bench_$C_.err
done
fi
VMAS_BENCH_SHARDED=1 python bench.py --no-cpu-baseline --no-other-configs --no-attached --steps 500 --warmup 50 > $OUT/${TAG}_bench_line_sharded_rollout_n1.json 2> $OUT/bench_sharded.err; tail -2 $OUT/bench_sharded.err
# rocprofv3 of the bench command itself (one queue: per-kernel durations, PMC passes, HBM traffic): the kernel trace runs
# behind bench.py's own clock warm-up; pmc_summary.py reports per-dispatch medians (whole run / last half)
export EVIDENCE_DIR=$TAG
BENCH="python $R/bench.py --no-cpu-baseline --no-fused --no-other-configs --no-attached --queues 1 --steps 2000 --warmup 200"
RATED=step_kernel_spec:physics bash scripts/gpu_counters.sh ${TAG}_bench_q1 384 1700 32768 -- $BENCH > /dev/null 2>&1
# ... and what bench.py itself measured in those profiled runs (the HIP events of a process under the profiler)
for p in trace p1; do grep -h '^{' /tmp/cnt_${TAG}_bench_q1/$p.log 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); print(json.dumps({'under':'rocprofv3 $p pass','ms_per_step':d['ms_per_step'],'frac':d['roofline']['frac'],'repeats':d['repeats']}))" ; done > $OUT/${TAG}_bench_q1_under_profiler.jsonl
$BENCH 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(json.dumps({'under':'no profiler, same command','ms_per_step':d['ms_per_step'],'frac':d['roofline']['frac'],'repeats':d['repeats']}))" >> $OUT/${TAG}_bench_q1_under_profiler.jsonl
cat $OUT/${TAG}_bench_q1_under_profiler.jsonl
# what one dependent launch costs on this box, and the shader clock with and without the profiler
{ echo "# scripts/micro/launch_floor (this round's box)"; scripts/micro/launch_floor 32768 8; echo "# under rocprofv3 --kernel-trace"; cd /tmp; export TMPDIR=/tmp; rocprofv3 --kernel-trace -d /tmp/lf -o lf -- $R/scripts/micro/launch_floor 32768 8 2>/dev/null | grep -E "clock probe|K0 empty  |K2 tile"; cd $R; } > $OUT/${TAG}_launch_floor.txt 2>&1
tail -12 $OUT/${TAG}_launch_floor.txt
# host side of Environment.step
python scripts/prof_env_host.py balance 32768 > $OUT/${TAG}_env_step_host_profile.txt 2>&1; head -12 $OUT/${TAG}_env_step_host_profile.txt
if [[ " $* " != *" quick "* && " $* " != *" no-shard-counters "* ]]; then
# counters of the latency-regime shards and the one-launch balance step (next-round items 2 and 3)
ACTIONS=zero RATED=step_kernel_spec_multi:env bash scripts/gpu_counters.sh ${TAG}_navigation8192_env_step 1480 30000 8192 -- python $S/bench_bound.py navigation 8192 > /dev/null 2>&1
COMPACT=1 FORCES=random RATED=step_kernel_compact:physics bash scripts/gpu_counters.sh ${TAG}_football16384_physics_compact 948 11900 16384 -- python $S/bench_world.py football 16384 300 > /dev/null 2>&1
RATED=step_kernel_spec_multi:env bash scripts/gpu_counters.sh ${TAG}_balance32768_env_step 657 2000 32768 -- python $S/bench_bound.py balance 32768 > /dev/null 2>&1
grep -h "sustained\|traffic / alg\|share of wave" $OUT/${TAG}_navigation8192_env_step_pmc_summary.txt $OUT/${TAG}_football16384_physics_compact_pmc_summary.txt $OUT/${TAG}_balance32768_env_step_pmc_summary.txt
fi
# rollout rates and per-phase traces
{ python $S/bench_rollout_env.py balance 32768 100; python $S/bench_rollout_env.py transport 16384 100; python $S/bench_rollout_env.py navigation 8192 50; REPS=5 python $S/bench_rollout_env.py football 131072 50; python $S/bench_rollout_env.py football 16384 50; } 2>&1 | grep "^{" > $OUT/${TAG}_env_rollout_rates.jsonl; cat $OUT/${TAG}_env_rollout_rates.jsonl
python $S/trace_compact.py football 16384 2>&1 | grep -v amdgpu > $OUT/${TAG}_football16384_compact_phase_trace.txt; cat $OUT/${TAG}_football16384_compact_phase_trace.txt
VMAS_TRACE=2 python $S/trace_nav.py 8192 2>&1 | grep -v amdgpu > $OUT/${TAG}_navigation8192_env_step_phase_trace.txt
python $S/trace_env.py balance 32768 2>&1 | grep -v amdgpu > $OUT/${TAG}_balance32768_env_step_phase_trace.txt; cat $OUT/${TAG}_balance32768_env_step_phase_trace.txt
[ -n "${FLAGS_AB:-}" ] || exit 0
# runtime flags A/B (kernel arguments in device memory, direct dispatch): the same driver-style command
for FL in "HIP_FORCE_DEV_KERNARG=0" "HIP_FORCE_DEV_KERNARG=1" "DEBUG_HIP_KERNARG_COPY_OPT=0" "GPU_MAX_HW_QUEUES=8"; do
  env $FL python bench.py --no-cpu-baseline --no-other-configs --no-attached --steps 2000 --warmup 200 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); e=d.get('environment_step',{}); print(json.dumps({'flag':'$FL','world_step_us':d['ms_per_step']*1e3,'env_step_us':e.get('us_per_step'),'env_step_gpu_us':e.get('gpu_us_per_step'),'bound_us':e.get('bound',{}).get('gpu_us_per_step'),'rollout_us':e.get('rollout',{}).get('us_per_step'),'persistent_us':d.get('persistent_rollout',{}).get('us_per_step')}))"
done > $OUT/${TAG}_runtime_flags_ab.jsonl
cat $OUT/${TAG}_runtime_flags_ab.jsonl
