# Round 4, GPU call t: the stand-alone football post-step with the gathered observation writer (one barrier, 3 KB of LDS)
# against its [64][D + 2] array (two barriers per agent), same library, same box; tests
TAG=r04t
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd $R
S=$R/scripts
timeout 1500 python -m pytest tests -m gpu -q --timeout=600 -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log
grep -E "^(FAILED|ERROR)|passed|failed|rc=|needed it|fixtures with any" $OUT/pytest_gpu.log | cut -c1-300 | head -20
grep -E "^E  +(Assertion|.*Error)" $OUT/pytest_gpu.log | cut -c1-300 | head -20
AB=$OUT/${TAG}_football_post_step_gathered_vs_shared_array.jsonl
: > $AB
for ROUND in 1 2; do for SH in 1 0; do
  export VMAS_FOOTBALL_POST_SHARED=$SH
  { for B in 32768 131072; do FOOTBALL_FORM=1 REPS=5 python $S/bench_rollout_env.py football $B 50; done
  } 2>&1 | grep "^{" | sed "s/^{/{\"post_step_writer\": \"$([ $SH = 1 ] && echo shared_array || echo gathered)\", /" >> $AB
  python bench.py --config football --no-cpu-baseline --no-attached --no-other-configs 2>> $OUT/bench.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); es=d['environment_step']
print(json.dumps({'post_step_writer': '$([ $SH = 1 ] && echo shared_array || echo gathered)', 'bench_py_environment_step_us': round(es['us_per_step'],2), 'gpu_us': round(es['gpu_us_per_step'],2)}))" >> $AB
done; done
unset VMAS_FOOTBALL_POST_SHARED
python - <<P
import json
for l in open("$AB"):
    r = json.loads(l)
    print(r["post_step_writer"].ljust(14), {k: v for k, v in r.items() if k in ("num_envs", "rollout_us_per_step_gpu", "step_us_per_step_wall", "bench_py_environment_step_us", "gpu_us")})
P
