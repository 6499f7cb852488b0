#!/bin/bash
# A measurement variant of the library that differs in the lane-compacted kernel's unit only:
#   scripts/variant_lib.sh NAME "-DVMAS_LZ_CUT=31"   ->  csrc/libvmas_hip_NAME.so (the product's other objects, this unit recompiled)
set -euo pipefail
cd "$(dirname "$0")/../vectorizedmultiagentsimulator_amd/csrc"
NAME=$1; EXTRA=${2:-}
HIPCC=/opt/rocm/bin/hipcc
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -Wall -Wno-unused-function "
COMMON="vmas_device.h vmas_env_device.h ../../include/vmas_hip.h ../../include/vmas_env_hip.h ../../include/vmas_debug_hip.h"
key() { (echo "$FLAGS"; "$HIPCC" --version | head -2; cat $1) | sha256sum | cut -c1-16; }
O1=.obj/vmas_hip.$(key "vmas_hip.hip vmas_step_device.h vmas_step_types.h vmas_spec_gen.h vmas_spec_kernel.h vmas_compact.h $COMMON").o
O2=.obj/vmas_env.$(key "vmas_env.hip $COMMON").o
[ -s "$O1" ] && [ -s "$O2" ] || { echo "build the product library first (bash build.sh)"; exit 1; }
mkdir -p /tmp/variant_obj
"$HIPCC" $FLAGS $EXTRA -c vmas_compact.hip -o /tmp/variant_obj/compact_$NAME.o
printf 'extern "C" { extern const char vmas_build_id_string[]; const char vmas_build_id_string[] = "variant-%s"; }\n' "$NAME" > /tmp/variant_obj/id_$NAME.cpp
g++ -O1 -fPIC -c /tmp/variant_obj/id_$NAME.cpp -o /tmp/variant_obj/id_$NAME.o
"$HIPCC" --offload-arch=gfx950 -fPIC -shared -Wl,-z,defs -o libvmas_hip_$NAME.so "$O1" "$O2" /tmp/variant_obj/compact_$NAME.o /tmp/variant_obj/id_$NAME.o
echo "built libvmas_hip_$NAME.so"
