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
srcs=(graph edge lin128 mlp128 mlp128s mlp128r node128 node128s pipeline train train_step heads)
newest_header=$(ls -t "$here"/*.h "$here"/../../include/*.h | head -1)
# No packed-fp32 instructions (v_pk_fma_f32, v_pk_mul_f32, v_pk_add_f32) in any kernel.  On MI355X a chain of dependent v_pk_*_f32
# instructions with op_sel modifiers returns a wrong LOW half for lanes 16..31 / 48..63 about once per 2e6 executions -- the value
# short of exactly one term of the chain -- for some alignments of the code (period 32 bytes) and only with two or more waves per
# SIMD; the same arithmetic as v_fma_f32 never fails (tools/probes/pk_chain_probe.hip + run_pad_sweep.sh: the stand-alone
# reproducer; profiles/r05_mol_fused2_soak.txt: how it was found, in round 5's molfuse2.hip -- removed in round 6).  Which kernels are exposed changes with every
# recompile, so the instruction class is off for all of them.  Cost: none (profiles/r05_no_packed_fp32_ab.txt).  The flag reaches
# the host pass too, which answers "not a recognized feature" once per file (filtered below).  A per-function
# target("no-packed-fp32-ops") attribute does the same job 10 % slower (the edge kernels lose their inlining): not used.
# NNHIP_PACKED_FP32=1 bash build.sh: the library with packed fp32, for A/B.
nopk=(-Xclang -target-feature -Xclang -packed-fp32-ops)
[ -n "${NNHIP_PACKED_FP32:-}" ] && nopk=()
jobs="${NNHIP_BUILD_JOBS:-8}"
pids=()
fail=0
running=0
for s in "${srcs[@]}"; do
  src="$here/$s.hip"
  obj="$objdir/$s.o"
  [ -f "$src" ] || continue
  if [ $force -eq 1 ] || [ ! -f "$obj" ] || [ "$src" -nt "$obj" ] || [ "$newest_header" -nt "$obj" ] || [ "${BASH_SOURCE[0]}" -nt "$obj" ]; then
    "$HIPCC" --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c "$src" -o "$obj" ${nopk[@]+"${nopk[@]}"} ${extra[@]+"${extra[@]}"} \
      2> >(grep -v "is not a recognized feature for this target" >&2) &
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
