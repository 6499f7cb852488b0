# Round 4, GPU call s: football's observations gathered from the tile as contiguous runs (football_obs_gather) in the step
# kernel's epilogue - against the r04q evidence build (libvmas_hip_prev.so: a row per lane / per-wave staging), same box, and
# the one-launch form against the two-launch form again by batch size; tests
TAG=r04s
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd $R
S=$R/scripts
timeout 1500 python -m pytest tests -m gpu -q --timeout=600 -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log
grep -E "^(FAILED|ERROR)|passed|failed|rc=|needed it|fixtures with any" $OUT/pytest_gpu.log | cut -c1-300 | head -20
grep -E "^E  +(Assertion|.*Error)" $OUT/pytest_gpu.log | cut -c1-300 | head -20
AB=$OUT/${TAG}_football_forms_gathered_observations.jsonl
: > $AB
for LIB in libvmas_hip_prev.so libvmas_hip.so; do
  export VMAS_HIP_LIB=$LIB
  { for B in 16384 32768 65536 131072; do for FORM in 0 1; do
      FOOTBALL_FORM=$FORM REPS=5 python $S/bench_rollout_env.py football $B 50
    done; done
  } 2>&1 | grep "^{" | sed "s/^{/{\"ab_library\": \"$LIB\", /" >> $AB
done
unset VMAS_HIP_LIB
python - <<P
import json
for l in open("$AB"):
    r = json.loads(l)
    print(r["ab_library"].ljust(20), r["num_envs"], "form", r["football_form"], {k: v for k, v in r.items() if k in ("rollout_us_per_step_gpu", "step_us_per_step_wall")})
P
for LIB in libvmas_hip_prev.so libvmas_hip.so; do
  VMAS_HIP_LIB=$LIB python bench.py --config football --no-cpu-baseline --no-attached --no-other-configs 2>> $OUT/bench.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); es=d['environment_step']
print('$LIB', 'physics', round(d['ms_per_step']*1e3,2), 'env step', round(es['us_per_step'],2), 'rollout', round(es['rollout']['us_per_step'],2), 'bound', round(es['bound']['us_per_step'],2))"
done
