#!/usr/bin/env bash
# Build libnewtonnet_hip.so for gfx950 (MI355X).  hipcc cross-compiles without a GPU present.
set -euo pipefail
here="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
out="$here/../lib"
mkdir -p "$out"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
"$HIPCC" --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC \
  "$here/graph.hip" "$here/edge.hip" "$here/lin128.hip" "$here/mlp128.hip" "$here/mlp128s.hip" "$here/node128.hip" "$here/node128s.hip" "$here/pipeline.hip" "$here/train.hip" "$here/train_step.hip" \
  -o "$out/${NNHIP_LIB_NAME:-libnewtonnet_hip.so}" "$@"
echo "built $out/${NNHIP_LIB_NAME:-libnewtonnet_hip.so}"
