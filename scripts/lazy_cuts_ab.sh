# World.step of the lane-compacted kernel (football, pinned): the lazy form against the per-environment form and the all-cut variant
cd ${GRAFT_REPO_ROOT:-.}
for NB in ${CUT_ENVS:-16384 8192 131072}; do
for R in 1 2; do
for L in libvmas_hip.so ${CUT_LIBS:-libvmas_hip_c31.so}; do
  [ -f vectorizedmultiagentsimulator_amd/csrc/$L ] || continue
  echo -n "football $NB $L exact=1: "; COMPACT=1 EXACT=1 FORCES=random QUEUES=1 VMAS_HIP_LIB=$L python scripts/bench_world.py football $NB 300 2>/dev/null | grep "^{" | python -c "import json,sys; print(json.loads(sys.stdin.readline())['world_step_us'])"
done
echo -n "football $NB libvmas_hip.so exact=0: "; COMPACT=1 EXACT=0 FORCES=random QUEUES=1 python scripts/bench_world.py football $NB 300 2>/dev/null | grep "^{" | python -c "import json,sys; print(json.loads(sys.stdin.readline())['world_step_us'])"
done; done
