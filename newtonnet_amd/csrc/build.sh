#!/usr/bin/env bash
# Build libnewtonnet_hip.so for gfx950 (MI355X).  hipcc cross-compiles without a GPU present.
#   build.sh            incremental: a source is recompiled when it (or any header) is newer than its object
#   build.sh --force    recompile every source (what __graft_entry__.build() runs)
#   build.sh -DFOO ...  tooling build: ANY extra compiler flag forces a full rebuild into its own object directory and marks
#                       the library (-DNNHIP_TOOLING: nnhip_build_flags() bit 0) -- the package refuses to load a marked library
#                       unless NNHIP_ALLOW_TOOLING_LIB=1, so a library built with an ablation switch can never serve results.
set -euo pipefail
here="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
out="$here/../lib"
mkdir -p "$out"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
force=0
extra=()
for a in "$@"; do
  if [ "$a" = "--force" ]; then force=1; else extra+=("$a"); fi
done
name="${NNHIP_LIB_NAME:-libnewtonnet_hip.so}"
objdir="$here/build/obj"
if [ ${#extra[@]} -gt 0 ]; then
  force=1
  objdir="$here/build/obj_${name%.so}_tooling"
  extra+=("-DNNHIP_TOOLING=1")
fi
mkdir -p "$objdir"
srcs=(graph edge lin128 mlp128 mlp128s mlp128r node128 node128s pipeline train train_step heads small molfuse molfuse2)
newest_header=$(ls -t "$here"/*.h "$here"/../../include/*.h | head -1)
jobs="${NNHIP_BUILD_JOBS:-8}"
pids=()
fail=0
running=0
for s in "${srcs[@]}"; do
  src="$here/$s.hip"
  obj="$objdir/$s.o"
  [ -f "$src" ] || continue
  if [ $force -eq 1 ] || [ ! -f "$obj" ] || [ "$src" -nt "$obj" ] || [ "$newest_header" -nt "$obj" ] || [ "${BASH_SOURCE[0]}" -nt "$obj" ]; then
    "$HIPCC" --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c "$src" -o "$obj" ${extra[@]+"${extra[@]}"} &
    pids+=($!)
    running=$((running + 1))
    if [ $running -ge "$jobs" ]; then
      wait "${pids[0]}" || fail=1
      pids=("${pids[@]:1}")
      running=$((running - 1))
    fi
  fi
done
for p in ${pids[@]+"${pids[@]}"}; do wait "$p" || fail=1; done
[ $fail -eq 0 ] || { echo "hipcc failed" >&2; exit 1; }
objs=()
for s in "${srcs[@]}"; do [ -f "$objdir/$s.o" ] && [ -f "$here/$s.hip" ] && objs+=("$objdir/$s.o"); done
"$HIPCC" --offload-arch=gfx950 -shared -fPIC "${objs[@]}" -o "$out/$name"
echo "built $out/$name"
