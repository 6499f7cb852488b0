timeout 900 python -m pytest tests/test_env_fused_gpu.py tests/test_env_gpu.py tests/test_scenarios_vs_reference.py -q --timeout=600 -p no:cacheprovider -m gpu -x -k "navigation or graph" 2>&1 | tail -3
export ACTIONS=zero
for B in 16384 32768 65536 131072; do timeout 120 python scripts/bench_bound.py navigation $B | tail -1; done
unset ACTIONS
ONLY=fused-eager python scripts/bench_env.py navigation 65536 | tail -1
ONLY=fused-graph python scripts/bench_env.py navigation 65536 | tail -1
OUT=gpurun_out/r4e; rm -rf $OUT; mkdir -p $OUT
ACTIONS=zero rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o env -- python scripts/bench_bound.py navigation 65536 > $OUT/stdout.log 2>&1
f=$(find $OUT/trace -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && grep "step_kernel\|collision" "$f" | cut -c1-200
rm -rf $OUT/trace
