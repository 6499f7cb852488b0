timeout 900 python -m pytest tests/test_env_fused_gpu.py tests/test_env_gpu.py tests/test_scenarios_vs_reference.py tests/test_hip_parity.py tests/test_specialize_gpu.py -q --timeout=600 -p no:cacheprovider -m gpu -x 2>&1 | tail -4
export ACTIONS=zero
for B in 8192 16384 65536; do timeout 120 python scripts/bench_bound.py navigation $B | tail -1; done
LANES=16 timeout 120 python scripts/bench_bound.py navigation 32768 | tail -1
LANES=8 timeout 120 python scripts/bench_bound.py navigation 32768 | tail -1
unset ACTIONS
for L in 8 16; do LANES=$L timeout 120 python scripts/bench_bound.py transport 32768 | tail -1; done
for L in 8 16; do LANES=$L timeout 120 python scripts/bench_bound.py transport 65536 | tail -1; done
timeout 120 python scripts/bench_bound.py balance 32768 | tail -1
for B in 8192 65536; do
  for T in 1 64; do echo "nav B=$B tiles=$T"; NAV_TILES=$T ONLY=fused-eager python scripts/bench_env.py navigation $B | tail -1; NAV_TILES=$T ONLY=fused-graph python scripts/bench_env.py navigation $B | tail -1; done
done
OUT=gpurun_out/r3x; rm -rf $OUT; mkdir -p $OUT
ACTIONS=zero rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o env -- python scripts/bench_bound.py navigation 65536 > $OUT/stdout.log 2>&1
f=$(find $OUT/trace -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cut -c1-200 "$f" | head -4
rm -rf $OUT/trace
