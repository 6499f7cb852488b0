export NAV_TILES=64 ACTIONS=zero
for L in 4 8 16; do LANES=$L timeout 120 python scripts/bench_bound.py navigation 65536 | tail -1; done
for L in 4 8 16; do LANES=$L timeout 120 python scripts/bench_bound.py navigation 32768 | tail -1; done
for L in 8 16; do LANES=$L timeout 120 python scripts/bench_bound.py navigation 16384 | tail -1; done
OUT=gpurun_out/r3w; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o env -- python scripts/bench_bound.py navigation 65536 > $OUT/stdout.log 2>&1
f=$(find $OUT/trace -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cut -c1-200 "$f" | head -4
rm -rf $OUT/trace
