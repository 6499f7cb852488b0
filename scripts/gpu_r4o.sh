TAG=r03
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
S=$R/scripts
{ for B in 8192 16384 32768 65536 131072; do ACTIONS=zero python scripts/bench_bound.py navigation $B; done; ACTIONS=zero python scripts/bench_bound.py navigation 16384; python scripts/bench_bound.py balance 32768; python scripts/bench_bound.py balance 65536; python scripts/bench_bound.py transport 16384; } 2>&1 | grep "^{" > $OUT/${TAG}_env_step_bound_rates.jsonl
cut -c1-330 $OUT/${TAG}_env_step_bound_rates.jsonl
export QUEUES=1 EVIDENCE_DIR=$TAG
ACTIONS=zero RATED=step_kernel_spec_multi bash scripts/gpu_counters.sh ${TAG}_navigation65536_env_step 1480 30000 65536 -- python $S/bench_bound.py navigation 65536 > /dev/null 2>&1
ACTIONS=zero RATED=step_kernel_spec_multi bash scripts/gpu_counters.sh ${TAG}_navigation8192_env_step 1480 30000 8192 -- python $S/bench_bound.py navigation 8192 > /dev/null 2>&1
unset QUEUES
python scripts/bench_rollout_env.py navigation 8192 50 2>&1 | grep "^{"
{ ONLY=fused-eager python scripts/bench_env.py navigation 65536; ONLY=fused-graph python scripts/bench_env.py navigation 65536; ONLY=fused-eager python scripts/bench_env.py navigation 8192; ONLY=fused-graph python scripts/bench_env.py navigation 8192; } 2>&1 | grep "^{" | cut -c1-250
VMAS_TRACE=2 NAV_TILES=64 python scripts/trace_nav.py 65536 > $OUT/${TAG}_navigation65536_env_step_phase_trace.txt 2>&1
VMAS_TRACE=2 NAV_TILES=64 python scripts/trace_nav.py 8192 > $OUT/${TAG}_navigation8192_env_step_phase_trace.txt 2>&1
grep "kernel trace\|achieved\|traffic /" $OUT/${TAG}_navigation65536_env_step_pmc_summary.txt $OUT/${TAG}_navigation8192_env_step_pmc_summary.txt | cut -c1-250
