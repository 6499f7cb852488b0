#!/bin/bash
# Build libvmas_hip.so for gfx950 (MI355X).  hipcc cross-compiles without a GPU.
#   -ffp-contract=off : reference operation order, no silent FMA fusion (parity)
set -euo pipefail
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
"$HIPCC" --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared \
  -Wall -Wno-unused-function \
  ${VMAS_HIPCC_EXTRA:-} \
  -o "${VMAS_LIB_OUT:-libvmas_hip.so}" vmas_hip.hip vmas_env.hip
echo "built $(pwd)/${VMAS_LIB_OUT:-libvmas_hip.so}"
