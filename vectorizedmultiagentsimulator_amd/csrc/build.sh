#!/bin/bash
# Build libvmas_hip.so for gfx950 (MI355X).  hipcc cross-compiles without a GPU.
#   -ffp-contract=off : reference operation order, no silent FMA fusion (parity)
# The three translation units compile side by side (the step kernel's template instantiations dominate), then link.
# Objects are cached under .obj/ (git-ignored) keyed by a hash of the unit's sources and flags: a change to the
# lane-compacted kernel recompiles in seconds instead of minutes.  VMAS_BUILD_FORCE=1 ignores the cache.
set -euo pipefail
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
OUT="${VMAS_LIB_OUT:-libvmas_hip.so}"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -Wall -Wno-unused-function ${VMAS_HIPCC_EXTRA:-}"
OBJ=.obj
mkdir -p "$OBJ"
COMMON="vmas_device.h vmas_env_device.h ../../include/vmas_hip.h ../../include/vmas_env_hip.h ../../include/vmas_debug_hip.h"
declare -A DEPS=(
  [vmas_hip]="vmas_hip.hip vmas_step_device.h vmas_step_types.h vmas_spec_gen.h vmas_spec_kernel.h vmas_compact.h $COMMON"
  [vmas_env]="vmas_env.hip $COMMON"
  [vmas_compact]="vmas_compact.hip vmas_compact.h vmas_step_types.h $COMMON"
)
pids=()
objs=()
for unit in vmas_hip vmas_env vmas_compact; do
  key=$( (echo "$FLAGS"; "$HIPCC" --version | head -2; cat ${DEPS[$unit]}) | sha256sum | cut -c1-16)
  o="$OBJ/$unit.$key.o"
  objs+=("$o")
  if [ -n "${VMAS_BUILD_FORCE:-}" ] || [ ! -s "$o" ]; then
    # (keep the three latest variants: product, profile, trace; `ls` fails on a fresh tree - no object yet - and must not end the script)
    (ls -t "$OBJ/$unit".*.o 2>/dev/null || true) | tail -n +4 | xargs -r rm -f
    ( "$HIPCC" $FLAGS -c "$unit.hip" -o "$o.tmp" && mv "$o.tmp" "$o" ) &
    pids+=($!)
  fi
done
for p in "${pids[@]:-}"; do [ -n "$p" ] && wait "$p"; done
# the build id (vmas_build_id): a digest of everything the library is built from, in a translation unit of its own - part of
# the key of the run-time specialisations' on-disk cache (specialize.py)
BUILD_ID=$( (echo "$FLAGS"; "$HIPCC" --version | head -2; cat vmas_hip.hip vmas_env.hip vmas_compact.hip *.h ../../include/*.h) | sha256sum | cut -c1-32)
printf 'extern "C" { extern const char vmas_build_id_string[]; const char vmas_build_id_string[] = "%s"; }\n' "$BUILD_ID" > "$OBJ/build_id.$BUILD_ID.cpp"
g++ -O1 -fPIC -c "$OBJ/build_id.$BUILD_ID.cpp" -o "$OBJ/build_id.$BUILD_ID.o"
# (-z defs: a symbol one unit declares and no unit defines fails HERE, not at dlopen on the GPU box)
"$HIPCC" --offload-arch=gfx950 -fPIC -shared -Wl,-z,defs -o "$OUT" "${objs[@]}" "$OBJ/build_id.$BUILD_ID.o"
rm -f "$OBJ/build_id.$BUILD_ID.cpp"
echo "built $(pwd)/$OUT"
