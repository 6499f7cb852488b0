# (the VMAS_ABLATE / VMAS_ENV_ABLATE knobs only exist in -DVMAS_PROFILE builds; the product library is rebuilt at the end)
VMAS_HIPCC_EXTRA=-DVMAS_PROFILE bash vectorizedmultiagentsimulator_amd/csrc/build.sh > /dev/null 2>&1
echo "--- ablations: 0 full | 1 no items | 16 descriptors only | 32 broad phase only | 2 no integrate | 3 neither"
for L in ${LANES:-8}; do for A in ${ABL:-0 1 16 32 2 3}; do VMAS_ABLATE=$A python bench.py --no-cpu-baseline --steps 1000 --warmup 100 --lanes $L ${BENCH_ARGS:-} 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('lanes $L ablate $A kernel_us %.2f'%d['roofline']['kernel_us'])"; done; done
bash vectorizedmultiagentsimulator_amd/csrc/build.sh > /dev/null 2>&1
