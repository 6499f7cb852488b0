# kernel-level profile of the fused Environment.step (eager), per scenario
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
for s in "balance 32768" "navigation 65536" "transport 16384"; do
  set -- $s
  rm -rf /tmp/prof_env
  ONLY=fused-eager rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_env -o env -- python $R/scripts/bench_env.py $1 $2 > /tmp/prof_env.log 2>&1
  f=$(find /tmp/prof_env -name "*kernel_stats.csv" | head -1)
  echo "== $1 $2"; grep scenario /tmp/prof_env.log
  cp "$f" $R/gpurun_out/env_${1}_kernel_stats.csv
  head -8 "$f" | cut -c1-200
done
