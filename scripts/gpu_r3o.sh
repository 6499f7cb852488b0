mkdir -p gpurun_out/r3o
for B in 8192 16384 65536; do
  for T in 1 64; do echo "tiles=$T"; NAV_TILES=$T python scripts/bench_bound.py navigation $B | tail -1; done
done
NAV_TILES=64 python scripts/trace_nav.py 65536 2>&1 | tail -14
NAV_TILES=64 python scripts/trace_nav.py 8192 2>&1 | tail -14
python scripts/bench_world.py navigation 65536 2>&1 | tail -2
