# Round-final evidence: bench line, rocprofv3 kernel trace + PMC for World.step, kernel stats of the fused
# Environment.step of the four benchmark scenarios, end-to-end rates.
mkdir -p gpurun_out/final
timeout 600 python bench.py > gpurun_out/final/bench.log 2>&1
grep '^{' gpurun_out/final/bench.log | cut -c1-2500
bash scripts/gpu_prof.sh > gpurun_out/final/prof_world_step.log 2>&1
grep -E "step_kernel|per-dispatch|bytes_per_launch" gpurun_out/final/prof_world_step.log | cut -c1-220 | head -30
cp gpurun_out/prof/latest_traffic.json gpurun_out/final/ 2>/dev/null
f=$(find gpurun_out/prof/trace -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" gpurun_out/final/world_step_kernel_stats.csv
R=$(pwd)
( cd /tmp && export TMPDIR=/tmp
for s in "balance 32768" "transport 16384" "navigation 65536" "football 131072"; do
  set -- $s
  rm -rf /tmp/prof_env
  ONLY=fused-eager rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_env -o env -- python $R/scripts/bench_env.py $1 $2 > /tmp/prof_env.log 2>&1
  f=$(find /tmp/prof_env -name "*kernel_stats.csv" | head -1)
  grep -v "at::\|rocclr" "$f" > $R/gpurun_out/final/env_${1}_${2}_kernel_stats.csv
  echo "== $1 $2 (under rocprofv3)"; grep scenario /tmp/prof_env.log; head -6 $R/gpurun_out/final/env_${1}_${2}_kernel_stats.csv | cut -c1-200
done )
echo "== end-to-end (no profiler)"
for s in "balance 32768" "transport 16384" "navigation 65536" "football 131072" "football 16384" "navigation 8192" "balance 1048576"; do
  for m in fused-eager fused-graph; do ONLY=$m python scripts/bench_env.py $s | grep scenario; done
done | tee gpurun_out/final/env_step_rates.jsonl
