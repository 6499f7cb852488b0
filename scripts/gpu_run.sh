# One parametrised GPU-call script (replaces the per-call gpu_r4*.sh of round 4):
#   gpurun -- 'TAG=r05a bash scripts/gpu_run.sh STAGE [STAGE...]'
# Everything lands in gpurun_out/$TAG/ (scratch); copy what is to be judged into profiles/.
# Stages:
#   tests-new        the attach(fused) tests + the tests touched this round (fast feedback)
#   tests            the whole -m gpu suite + smoke()
#   bench            bench.py default (every leg) + the driver-style invocation
#   bench-configs    the full line of every other configuration
#   attached         only the attached_reference legs of all five configurations
#   ab LIB [LIB..] -- CMD...   A/B of libraries: CMD (a scripts/bench_*.py line printer) under each VMAS_HIP_LIB, twice, interleaved
#   bw               bandwidth regime: balance at 262144 and 1048576 environments with kernel trace + counters
#   counters-shards  counters of the latency-regime shards (navigation 8192 env step, football 16384 compact, balance env step)
#   evidence         rocprofv3 of the bench command itself (one queue): kernel stats + PMC summary
#   traces           per-phase traces (football compact, navigation env step, balance env step)
#   lazy-parts       round 6: the lazy form's parts in the compacted kernel (profiling library)
#   lazy / lazy-cost round 6: the lazy exact broad phase's tests / its cost against the per-environment form and round 5's library
TAG=${TAG:-r05}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd $R
S=$R/scripts
export EVIDENCE_DIR=$TAG

show_pytest() {  # the interesting lines of a pytest log
  grep -E "^(FAILED|ERROR)|passed|failed|rc=|needed it|fixtures with any" "$1" | cut -c1-400 | head -40
  grep -E "^E  +" "$1" | cut -c1-400 | head -60
}

while [ $# -gt 0 ]; do
  STAGE=$1; shift
  case $STAGE in
  tests-new)
    timeout 1500 python -m pytest tests/test_attached_env_gpu.py tests/test_adapter_reference.py tests/test_output_pool.py \
      tests/test_env_fused_gpu.py tests/test_round4_gpu.py -m gpu -q --timeout=600 -p no:cacheprovider ${PYTEST_X:-} --tb=short > $OUT/pytest_new.log 2>&1
    echo "pytest rc=$?" >> $OUT/pytest_new.log
    show_pytest $OUT/pytest_new.log; tail -n 40 $OUT/pytest_new.log | cut -c1-300
    ;;
  tests)
    rm -f gpurun_out/parity_allowance.jsonl
    timeout 2400 python -m pytest tests -m gpu -q --timeout=600 -p no:cacheprovider -rs > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log
    show_pytest $OUT/pytest_gpu.log
    cp gpurun_out/parity_allowance.jsonl $OUT/${TAG}_parity_allowance.jsonl 2>/dev/null
    cp gpurun_out/broad_phase_lazy.jsonl $OUT/${TAG}_broad_phase_lazy.jsonl 2>/dev/null
    python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
    ;;
  bench)
    ( time python bench.py > $OUT/${TAG}_bench_line_default.json 2> $OUT/bench_default.err ) 2>&1 | grep real
    tail -3 $OUT/bench_default.err
    python $S/show_line.py $OUT/${TAG}_bench_line_default.json
    python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $OUT/${TAG}_bench_line_driver_style.json 2> $OUT/bench_driver_style.err
    python $S/show_line.py $OUT/${TAG}_bench_line_driver_style.json brief
    ;;
  bench-configs)
    for C_ in transport transport_2pkg navigation football; do
      python bench.py --config $C_ > $OUT/${TAG}_bench_line_$C_.json 2> $OUT/bench_$C_.err; python $S/show_line.py $OUT/${TAG}_bench_line_$C_.json brief; tail -2 $OUT/bench_$C_.err
    done
    ;;
  attached)
    python $S/bench_attached.py > $OUT/${TAG}_attached_reference.jsonl 2> $OUT/attached.err; cat $OUT/${TAG}_attached_reference.jsonl | cut -c1-900; tail -5 $OUT/attached.err
    ;;
  ab)
    LIBS=(); while [ $# -gt 0 ] && [ "$1" != "--" ]; do LIBS+=("$1"); shift; done; shift
    CMD="$*"; set --
    NAME=$(echo "${LIBS[*]:1}" | tr ' /' '__' | sed 's/libvmas_hip_//g; s/\.so//g')
    AB=$OUT/${TAG}_ab_${NAME}.jsonl; : > $AB
    for ROUND in 1 2; do for LIB in "${LIBS[@]}"; do
      VMAS_HIP_LIB=$LIB bash -c "$CMD" 2>&1 | grep "^{" | sed "s/^{/{\"ab_library\": \"$LIB\", /" >> $AB
    done; done
    cut -c1-300 $AB
    ;;
  bw)
    for B_ in 262144 1048576; do
      { FORCES=random QUEUES=1 python $S/bench_world.py balance $B_ 300; FORCES=random python $S/bench_world.py balance $B_ 300; } 2>&1 | grep "^{" | tee -a $OUT/${TAG}_bandwidth_regime.jsonl
      FORCES=random QUEUES=1 RATED=step_kernel_spec:physics bash $S/gpu_counters.sh ${TAG}_balance${B_}_physics 384 1700 $B_ -- python $S/bench_world.py balance $B_ 300 > /dev/null 2>&1
      grep -h "sustained\|traffic / alg\|share of wave\|median" $OUT/${TAG}_balance${B_}_physics_pmc_summary.txt | head -12
    done
    scripts/micro/launch_floor 1048576 8 2>&1 | tail -8 | tee $OUT/${TAG}_launch_floor_1M.txt
    ;;
  counters-shards)
    ACTIONS=zero RATED=step_kernel_spec_multi:env bash $S/gpu_counters.sh ${TAG}_navigation8192_env_step 1480 30000 8192 -- python $S/bench_bound.py navigation 8192 > /dev/null 2>&1
    COMPACT=1 FORCES=random RATED=step_kernel_compact:physics bash $S/gpu_counters.sh ${TAG}_football16384_physics_compact 948 11900 16384 -- python $S/bench_world.py football 16384 300 > /dev/null 2>&1
    RATED=step_kernel_spec_multi:env bash $S/gpu_counters.sh ${TAG}_balance32768_env_step 657 2000 32768 -- python $S/bench_bound.py balance 32768 > /dev/null 2>&1
    grep -h "sustained\|traffic / alg\|share of wave" $OUT/${TAG}_navigation8192_env_step_pmc_summary.txt $OUT/${TAG}_football16384_physics_compact_pmc_summary.txt $OUT/${TAG}_balance32768_env_step_pmc_summary.txt
    ;;
  evidence)
    # the bench command itself under rocprofv3.  (1) the HEADLINE of round 6: env.step of the attached reference environment - its
    # one-launch kernel rated at the physics' bytes + the actions / observations / rewards / done / info (657 B per environment)
    BENCH="python $R/bench.py --no-cpu-baseline --no-fused --no-other-configs --no-traffic --steps 2000 --warmup 200 --repeats 3"
    RATED=step_kernel_spec_multi:env bash $S/gpu_counters.sh ${TAG}_bench_q1 657 1700 32768 -- $BENCH > /dev/null 2>&1
    grep -h "sustained\|traffic / alg\|share of wave\|median" $OUT/${TAG}_bench_q1_pmc_summary.txt | head
    # (2) the physics-only launches (`world_step`), as in rounds 1-5
    BENCHW="$BENCH --no-attached --queues 1"
    RATED=step_kernel_spec:physics bash $S/gpu_counters.sh ${TAG}_bench_world_step_q1 384 1700 32768 -- $BENCHW > /dev/null 2>&1
    grep -h "sustained\|traffic / alg\|share of wave\|median" $OUT/${TAG}_bench_world_step_q1_pmc_summary.txt | head
    { echo "# scripts/micro/launch_floor (this round's box)"; scripts/micro/launch_floor 32768 8; } > $OUT/${TAG}_launch_floor.txt 2>&1; tail -6 $OUT/${TAG}_launch_floor.txt
    ;;
  traces)
    python $S/trace_compact.py football 16384 2>&1 | grep -v amdgpu > $OUT/${TAG}_football16384_compact_phase_trace.txt; tail -20 $OUT/${TAG}_football16384_compact_phase_trace.txt
    VMAS_TRACE=2 python $S/trace_nav.py 8192 2>&1 | grep -v amdgpu > $OUT/${TAG}_navigation8192_env_step_phase_trace.txt; tail -20 $OUT/${TAG}_navigation8192_env_step_phase_trace.txt
    ;;
  lazy)   # round 6: the lazy exact broad phase - its own tests, the exact-form tests it now serves, and what it costs
    rm -f gpurun_out/broad_phase_lazy.jsonl
    timeout 2400 python -m pytest tests/test_broad_phase_lazy_gpu.py tests/test_hip_parity.py tests/test_compact_gpu.py tests/test_specialize_gpu.py \
      -m gpu -q --timeout=900 -p no:cacheprovider ${PYTEST_X:-} --tb=short -k "${PYTEST_K:-exact or band or lazy or bitwise or reference_rule or waits or gated}" > $OUT/pytest_lazy.log 2>&1
    echo "pytest rc=$?" >> $OUT/pytest_lazy.log
    show_pytest $OUT/pytest_lazy.log; tail -n 15 $OUT/pytest_lazy.log | cut -c1-300
    cp gpurun_out/broad_phase_lazy.jsonl $OUT/${TAG}_broad_phase_lazy.jsonl 2>/dev/null; cat $OUT/${TAG}_broad_phase_lazy.jsonl 2>/dev/null | cut -c1-400
    ;;
  lazy-cost)   # World.step with the reference's rule (lazy form) against the per-environment form and the previous round's library
    AB=$OUT/${TAG}_lazy_cost.jsonl; : > $AB
    for ROUND in 1 2; do
      for CFG in ${LAZY_CFGS:-balance:32768 transport:16384 football:16384 football:8192 football:131072 balance:1048576}; do
        SC=${CFG%%:*}; NB=${CFG##*:}
        PIN=; [ $SC = football ] && PIN=1   # (football: the lane-compacted kernel pinned - the library's choice alternates with the interpreter)
        for E in 1 0; do COMPACT=$PIN EXACT=$E FORCES=random QUEUES=1 python $S/bench_world.py $SC $NB 300 2>$OUT/lazy_cost.err | grep "^{" >> $AB; done
        if [ -f vectorizedmultiagentsimulator_amd/csrc/libvmas_hip_r5.so ]; then
          COMPACT=$PIN EXACT=0 FORCES=random QUEUES=1 VMAS_HIP_LIB=libvmas_hip_r5.so python $S/bench_world.py $SC $NB 300 2>$OUT/lazy_cost_r5.err | grep "^{" >> $AB || tail -3 $OUT/lazy_cost_r5.err
        fi
      done
    done
    python - "$AB" <<'PY'
import json, sys
rows = [json.loads(l) for l in open(sys.argv[1])]
for r in rows:
    print(f"{r['scenario']:10s} {r['num_envs']:8d} lib={r['lib'][-16:]:16s} exact={r.get('exact')} form={r.get('exact_form')} {r['world_step_us']:8.2f} us  spec={r['specialized']} compact={r['compact']}")
PY
    ;;
  lazy-parts)  # what each part of the lazy form costs the compacted kernel (profiling library: VMAS_ABLATE bits 8 / 10)
    LP=$OUT/${TAG}_lazy_parts.txt; : > $LP
    for CFG in ${LAZY_CFGS:-football:16384 football:131072}; do
      SC=${CFG%%:*}; NB=${CFG##*:}
      for ROUND in 1 2; do
        for V in "1:0:lazy form (product)" "1:1024:no band flags -> no tile asks" "1:1280:... and no overlap tests in the broad phase" "0:0:per-environment form"; do
          E=${V%%:*}; R=${V#*:}; AB=${R%%:*}; WHAT=${R#*:}
          echo -n "$SC $NB | $WHAT | " >> $LP
          COMPACT=1 EXACT=$E VMAS_ABLATE=$AB FORCES=random QUEUES=1 VMAS_HIP_LIB=libvmas_hip_prof.so python $S/bench_world.py $SC $NB 300 2>$OUT/lazy_parts.err | grep "^{" | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['world_step_us'], 'us')" >> $LP || tail -3 $OUT/lazy_parts.err
        done
      done
    done
    cat $LP
    ;;
  lazy-counters)  # instruction counts / instruction-cache behaviour of the compacted kernel with and without the lazy form
    LC=$OUT/${TAG}_lazy_counters.txt; : > $LC
    cd /tmp && export TMPDIR=/tmp
    for E in 0 1; do
      for SET in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY" \
                 "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_ANY SQ_WAIT_ANY"; do
        W=/tmp/lc_$E; rm -rf $W
        COMPACT=1 EXACT=$E FORCES=random QUEUES=1 rocprofv3 --pmc $SET --output-format csv -d $W -o p -- python $S/bench_world.py football ${LC_ENVS:-16384} 200 > $W.log 2>&1
        python - $W $E >> $LC <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "step_kernel_compact" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
print(f"exact={sys.argv[2]}", {k: round(sorted(v)[len(v) // 2]) for k, v in sorted(acc.items())}, "dispatches", max((len(v) for v in acc.values()), default=0))
PY
      done
    done
    cd $R; cat $LC
    ;;
  *) echo "unknown stage $STAGE";;
  esac
done
