# after the closing call: navigation's counters on the final build, and the GPU suite once more
TAG=r03
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
S=$R/scripts
export QUEUES=1 EVIDENCE_DIR=$TAG
ACTIONS=zero RATED=step_kernel_spec_multi bash scripts/gpu_counters.sh ${TAG}_navigation65536_env_step 1480 30000 65536 -- python $S/bench_bound.py navigation 65536 > /dev/null 2>&1
ACTIONS=zero RATED=step_kernel_spec_multi bash scripts/gpu_counters.sh ${TAG}_navigation8192_env_step 1480 30000 8192 -- python $S/bench_bound.py navigation 8192 > /dev/null 2>&1
unset QUEUES
grep "kernel trace\|achieved\|traffic /" $OUT/${TAG}_navigation65536_env_step_pmc_summary.txt $OUT/${TAG}_navigation8192_env_step_pmc_summary.txt | cut -c60-250
VMAS_TRACE=2 python scripts/trace_nav.py 65536 2>&1 | grep -v amdgpu > $OUT/${TAG}_navigation65536_env_step_phase_trace.txt
VMAS_TRACE=2 python scripts/trace_nav.py 8192 2>&1 | grep -v amdgpu > $OUT/${TAG}_navigation8192_env_step_phase_trace.txt
for H in 0 150 600; do HOLD=$H python scripts/trace_compact.py football 131072 2>&1 | grep -v amdgpu | tail -11; done > $OUT/${TAG}_football131072_compact_phases_by_contact_density.txt
timeout 1800 python -m pytest tests -m gpu -q --timeout=600 -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log
grep -E "^(FAILED|ERROR)|passed|failed|rc=" $OUT/pytest_gpu.log | cut -c1-300 | head
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
