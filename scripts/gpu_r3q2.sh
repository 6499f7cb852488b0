export NAV_TILES=64
for B in 65536 8192; do
OUT=gpurun_out/r3q_$B; rm -rf $OUT; mkdir -p $OUT
ACTIONS=zero rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o env -- python scripts/bench_bound.py navigation $B > $OUT/stdout.log 2>&1
f=$(find $OUT/trace -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cut -c1-200 "$f" | head -8
tail -2 $OUT/stdout.log
rm -rf $OUT/trace
done
