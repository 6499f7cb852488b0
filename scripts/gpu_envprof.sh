# per-kernel time of one Environment.step (graph replay) of a scenario: bash scripts/gpu_envprof.sh navigation 8192
set -u
N=${1:-navigation}; B=${2:-8192}
OUT=gpurun_out/envprof_${N}_${B}; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
ONLY=${ONLY:-fused-eager} rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o env -- python scripts/bench_env.py $N $B > $OUT/stdout.log 2>&1
f=$(find $OUT/trace -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cut -c1-260 "$f" | head -14 | tee $OUT/kernel_stats_head.csv
grep "^{" $OUT/stdout.log
rm -rf $OUT/trace
