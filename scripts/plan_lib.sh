#!/bin/bash
# A host-only PLANNING build of the library (seconds, no device code): vmas_world_create(device -1), the planners and the
# debug dumps work, nothing can be launched.  What scripts/gen_spec.py generates csrc/vmas_spec_gen.h with - it does not
# include that header (-DVMAS_PLAN_ONLY), so a change of the planner never needs the stale header to compile.
#   bash scripts/plan_lib.sh [out.so]   (default: vectorizedmultiagentsimulator_amd/csrc/libvmas_plan.so)
set -euo pipefail
cd "$(dirname "$0")/../vectorizedmultiagentsimulator_amd/csrc"
OUT=${1:-libvmas_plan.so}
TMP=$(mktemp -d)
trap 'rm -rf "$TMP"' EXIT
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
for u in vmas_hip vmas_env vmas_compact; do
  timeout 300 "$HIPCC" --offload-arch=gfx950 --cuda-host-only -DVMAS_PLAN_ONLY -O1 -std=c++17 -ffp-contract=off -fPIC -Wno-unused-function -c $u.hip -o "$TMP/$u.o"
done
# (a host-only object still refers to its fat binary: an empty one will do - no kernel is ever launched from this library)
{ echo 'extern "C" { extern const char vmas_build_id_string[]; const char vmas_build_id_string[] = "plan-only";'
  for s in $(nm -u "$TMP"/*.o | grep -o "__hip_fatbin_[0-9a-f]*" | sort -u); do echo "char $s[256] __attribute__((aligned(4096))) = {0};"; done
  echo '}'; } > "$TMP/stub.cpp"
g++ -fPIC -c "$TMP/stub.cpp" -o "$TMP/stub.o"
"$HIPCC" -fPIC -shared -o "$OUT" "$TMP"/vmas_hip.o "$TMP"/vmas_env.o "$TMP"/vmas_compact.o "$TMP/stub.o"
echo "built $(pwd)/$OUT (planning only)"
