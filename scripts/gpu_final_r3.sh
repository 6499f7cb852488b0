# The round's closing GPU call: the whole GPU suite, smoke, and the rates the last kernel changes moved (-> gpurun_out/r03/)
TAG=r03
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
rm -f gpurun_out/parity_allowance.jsonl
timeout 1800 python -m pytest tests -m gpu -q --timeout=600 -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log
grep -E "^(FAILED|ERROR)|passed|failed|rc=" $OUT/pytest_gpu.log | cut -c1-300 | head -20
cp gpurun_out/parity_allowance.jsonl $OUT/${TAG}_parity_allowance.jsonl 2>/dev/null
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
{ for B in 8192 16384 32768 65536 131072; do ACTIONS=zero python scripts/bench_bound.py navigation $B; done; python scripts/bench_bound.py balance 32768; python scripts/bench_bound.py balance 65536; python scripts/bench_bound.py transport 16384; } 2>&1 | grep "^{" > $OUT/${TAG}_env_step_bound_rates.jsonl
{ python scripts/bench_rollout_env.py balance 32768 100; python scripts/bench_rollout_env.py transport 16384 100; python scripts/bench_rollout_env.py navigation 8192 50; REPS=5 python scripts/bench_rollout_env.py football 131072 50; python scripts/bench_rollout_env.py football 16384 50; } 2>&1 | grep "^{" > $OUT/${TAG}_env_rollout_rates.jsonl
{ for W in "balance 32768" "transport 16384" "navigation 65536" "navigation 8192" "football 131072" "football 16384"; do ONLY=fused-eager python scripts/bench_env.py $W; ONLY=fused-graph python scripts/bench_env.py $W; done; } 2>&1 | grep "^{" > $OUT/${TAG}_env_step_rates.jsonl
# the library's choice of kernel for football by contact density (one queue, 600 steps): auto / compacted / interpreter
{ for F in random fixed; do for CP in "" 1 0; do QUEUES=1 FORCES=$F COMPACT=$CP python scripts/bench_world.py football 131072 600 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); d['compact_mode']='${CP:-auto}'; print(json.dumps(d))"; done; done; } > $OUT/${TAG}_football_kernel_choice.jsonl
python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $OUT/${TAG}_bench_line_driver_style.json 2>/dev/null
python bench.py --no-cpu-baseline > $OUT/${TAG}_bench_line_default_nocpu.json 2>/dev/null
cut -c1-260 $OUT/${TAG}_env_step_bound_rates.jsonl; cut -c1-200 $OUT/${TAG}_env_rollout_rates.jsonl; cut -c1-230 $OUT/${TAG}_football_kernel_choice.jsonl
