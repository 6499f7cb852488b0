# time bench.py's kernel with each library in csrc/variants/*.so (copied over the product library one at a time)
L=vectorizedmultiagentsimulator_amd/csrc
cp $L/libvmas_hip.so /tmp/orig.so
for rep in 1 2; do for v in ${VARIANTS:-$(ls $L/variants/*.so)}; do
  cp $v $L/libvmas_hip.so
  python bench.py --no-cpu-baseline --no-fused --steps 3000 --warmup 300 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$(basename $v) balance kernel_us %.2f'%d['roofline']['kernel_us'])"
  [ -n "$MORE" ] && for s in "transport 16384" "football 16384" "navigation 65536"; do echo "$(basename $v) $(python scripts/bench_world.py $s 2>/dev/null | tail -1 | cut -c1-120)"; done
done; done
cp /tmp/orig.so $L/libvmas_hip.so
