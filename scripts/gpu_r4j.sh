# Round 4, tenth GPU call: football's three forms (one launch | two per step | two per step with the post-steps on a second
# queue) by batch size, rollout outputs preallocated (scripts/bench_rollout_env.py); tests; the football bench line
TAG=r04j
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd $R
S=$R/scripts
timeout 1500 python -m pytest tests -m gpu -q --timeout=600 -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log
grep -E "^(FAILED|ERROR)|passed|failed|rc=|needed it|fixtures with any" $OUT/pytest_gpu.log | cut -c1-300 | head -20
grep -E "^E  +(Assertion|.*Error)" $OUT/pytest_gpu.log | cut -c1-300 | head -20
{ for B in 16384 32768 65536 131072; do for FORM in 0 1 2; do
    FOOTBALL_FORM=$FORM REPS=5 python $S/bench_rollout_env.py football $B 50 2>&1 | grep "^{"
  done; done; } > $OUT/${TAG}_football_forms_by_batch.jsonl; cut -c1-330 $OUT/${TAG}_football_forms_by_batch.jsonl
{ for B in 8192 16384 32768 131072; do REPS=5 python $S/bench_rollout_env.py football $B 50 2>&1 | grep "^{"; done; } > $OUT/${TAG}_football_library_choice.jsonl; cut -c1-330 $OUT/${TAG}_football_library_choice.jsonl
python bench.py --config football --no-cpu-baseline --no-attached > $OUT/${TAG}_bench_line_football.json 2>> $OUT/bench.err
python - <<P
import json
d = json.loads(open("$OUT/${TAG}_bench_line_football.json").read().strip().splitlines()[-1])
print({k: d.get(k) for k in ("value", "ms_per_step")}, {k: (v if not isinstance(v, dict) else {kk: vv for kk, vv in v.items() if "us" in kk or "value" in kk}) for k, v in d.items() if k in ("environment_step", "persistent_rollout", "sharded_rollout")})
P
VMAS_BENCH_SHARDED=1 python bench.py --config football --no-cpu-baseline --no-attached --no-other-configs > $OUT/${TAG}_bench_line_football_sharded_n1.json 2>> $OUT/bench.err; python - <<P
import json
d = json.loads(open("$OUT/${TAG}_bench_line_football_sharded_n1.json").read().strip().splitlines()[-1])
print({k: v for k, v in d.items() if "shard" in k or "gather" in k})
P
