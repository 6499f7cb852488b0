cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for B in 16384 131072; do
  rm -rf /tmp/prof_env
  ONLY=fused-eager rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_env -o env -- python $R/scripts/bench_env.py football $B > /tmp/prof_env.log 2>&1
  f=$(find /tmp/prof_env -name "*kernel_stats.csv" | head -1)
  echo "== football $B"; grep scenario /tmp/prof_env.log; grep "kernel" "$f" | grep -v "at::" | cut -c1-190 | head -4
done
ONLY=plain-graph python $R/scripts/bench_env.py football 16384 | grep scenario
