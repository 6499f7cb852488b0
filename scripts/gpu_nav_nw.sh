cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for nw in 4 8; do
  rm -rf /tmp/prof_env
  VMAS_NAV_NW=$nw ONLY=fused-eager rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_env -o env -- python $R/scripts/bench_env.py navigation 65536 > /tmp/prof_env.log 2>&1
  f=$(find /tmp/prof_env -name "*kernel_stats.csv" | head -1)
  echo "== nw=$nw"; grep scenario /tmp/prof_env.log; grep "navigation_post" "$f" | cut -c1-200
done
