#!/bin/bash
# Build libvmas_hip.so for gfx950 (MI355X).  hipcc cross-compiles without a GPU.
#   -ffp-contract=off : reference operation order, no silent FMA fusion (parity)
# The two translation units compile side by side (the step kernel's template instantiations dominate), then link.
set -euo pipefail
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
OUT="${VMAS_LIB_OUT:-libvmas_hip.so}"
OBJ=$(mktemp -d)
trap 'rm -rf "$OBJ"' EXIT
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -Wall -Wno-unused-function ${VMAS_HIPCC_EXTRA:-}"
"$HIPCC" $FLAGS -c vmas_hip.hip -o "$OBJ/vmas_hip.o" &
p1=$!
"$HIPCC" $FLAGS -c vmas_env.hip -o "$OBJ/vmas_env.o" &
p2=$!
wait $p1
wait $p2
"$HIPCC" --offload-arch=gfx950 -fPIC -shared -o "$OUT" "$OBJ/vmas_hip.o" "$OBJ/vmas_env.o"
echo "built $(pwd)/$OUT"
