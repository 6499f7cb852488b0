# An A/B prepared in round 4 and NOT yet measured: broad-phase units of up to 12 partners, tested
# six at a time against one fetch of the row entity (csrc/vmas_compact.h, VMAS_COMPACT_UNIT_CHUNKS; the default build is
# byte for byte without it).  Before the GPU call, on the CPU side:
#   cd vectorizedmultiagentsimulator_amd/csrc && VMAS_LIB_OUT=libvmas_hip_chunks2.so VMAS_HIPCC_EXTRA="-DVMAS_COMPACT_UNIT_CHUNKS=2" bash build.sh
# (football 5 v 5 at 16 waves per tile: 39 units instead of 50, ten of them a line against ten spheres; registers 73 / 79 for
#  the physics forms, 82 - one wave per SIMD less - for the ingest form with two owned entities per wave).
# Then:  gpurun -- 'bash scripts/gpu_next_unit_chunks.sh'
TAG=${TAG:-r05_unit_chunks}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd $R
S=$R/scripts
# bitwise against the interpreter first (the suite's compacted-kernel tests on the experimental library)
VMAS_HIP_LIB=libvmas_hip_chunks2.so timeout 600 python -m pytest tests/test_compact_gpu.py tests/test_round4_gpu.py -m gpu -q -p no:cacheprovider 2>&1 | tail -n 3
AB=$OUT/${TAG}_ab.jsonl
: > $AB
for LIB in libvmas_hip.so libvmas_hip_chunks2.so libvmas_hip.so libvmas_hip_chunks2.so; do
  export VMAS_HIP_LIB=$LIB
  { COMPACT=1 FORCES=random QUEUES=1 python $S/bench_world.py football 131072 300
    COMPACT=1 FORCES=random QUEUES=2 python $S/bench_world.py football 131072 300
    COMPACT=1 FORCES=random python $S/bench_world.py football 16384 300
  } 2>&1 | grep "^{" | sed "s/^{/{\"ab_library\": \"$LIB\", /" >> $AB
done
unset VMAS_HIP_LIB
cut -c1-260 $AB
