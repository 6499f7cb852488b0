# Round 4, GPU call r: the world-specialised kernel's load phase through scalar bases + the 32-bit lane (no 64-bit per-lane
# offsets: navigation's 16-wave kernel spilled one and waited for every load in front of its reload) and navigation's
# prologue tables requested in one burst - against the r04q evidence build (libvmas_hip_prev.so), same box; tests
TAG=r04r
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd $R
S=$R/scripts
timeout 1500 python -m pytest tests -m gpu -q --timeout=600 -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log
grep -E "^(FAILED|ERROR)|passed|failed|rc=|needed it|fixtures with any" $OUT/pytest_gpu.log | cut -c1-300 | head -20
grep -E "^E  +(Assertion|.*Error)" $OUT/pytest_gpu.log | cut -c1-300 | head -20
AB=$OUT/${TAG}_ab_evidence_build_vs_this.jsonl
: > $AB
for LIB in libvmas_hip_prev.so libvmas_hip.so libvmas_hip_prev.so libvmas_hip.so; do
  export VMAS_HIP_LIB=$LIB
  { ACTIONS=zero python $S/bench_bound.py navigation 8192
    ACTIONS=zero python $S/bench_bound.py navigation 65536
    python $S/bench_bound.py balance 32768
    python $S/bench_bound.py transport 16384
    python $S/bench_rollout_env.py navigation 8192 50
    python $S/bench_rollout_env.py balance 32768 100
  } 2>&1 | grep "^{" | sed "s/^{/{\"ab_library\": \"$LIB\", /" >> $AB
done
unset VMAS_HIP_LIB
python - <<P
import json
for l in open("$AB"):
    r = json.loads(l)
    print(r["ab_library"].ljust(20), r["scenario"], r["num_envs"], r.get("specialized"), {k: v for k, v in r.items() if k in ("step_bound_us", "rollout_us_per_step_gpu", "step_us_per_step_wall")})
P
VMAS_TRACE=2 python $S/trace_nav.py 8192 2>&1 | grep -v amdgpu > $OUT/${TAG}_navigation8192_env_step_phase_trace.txt; head -n 14 $OUT/${TAG}_navigation8192_env_step_phase_trace.txt
COMPACT=1 FORCES=random RATED=step_kernel_compact:physics EVIDENCE_DIR=$TAG bash scripts/gpu_counters.sh ${TAG}_football16384_physics_compact 948 11900 16384 -- python $S/bench_world.py football 16384 300 > /dev/null 2>&1
grep -h "per dispatch\|sustained\|traffic / alg" $OUT/${TAG}_football16384_physics_compact_pmc_summary.txt | head -6
