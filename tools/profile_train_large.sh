#!/usr/bin/env bash
# rocprofv3 evidence for the large-batch training roofline (bench.py:train_roofline): kernel trace + FETCH_SIZE / WRITE_SIZE passes
# of tools/train_large_pass.py (1024 aspirin conformers, value + tangent sweeps + weight gradients, no optimizer, one rank).
# usage (through gpurun): tools/profile_train_large.sh <tag>
set -uo pipefail
tag="${1:-r03}"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out
sha="$(cat profiles/.tree_sha 2>/dev/null || echo unknown)"
rocprofv3 --kernel-trace --stats -d $out/${tag}_tl_trace -o t -- python3 tools/train_large_pass.py 5 > $out/${tag}_tl_trace.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $out/${tag}_tl_fetch -o f -- python3 tools/train_large_pass.py 2 > $out/${tag}_tl_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $out/${tag}_tl_write -o w -- python3 tools/train_large_pass.py 2 > $out/${tag}_tl_write.log 2>&1
{ echo "# tree $sha"; echo "# cmd python3 tools/train_large_pass.py 5 (1024 aspirin conformers: value + tangent sweeps + weight gradients)"; python3 tools/rocpd_stats.py $out/${tag}_tl_trace/t_results.db --by-grid; } > $out/${tag}_train_aspirin1024_kernel_stats.txt
{ echo "# tree $sha"; python3 tools/rocpd_pmc.py $out/${tag}_tl_fetch/f_results.db; } > $out/${tag}_train_aspirin1024_pmc_fetch_size.txt
{ echo "# tree $sha"; python3 tools/rocpd_pmc.py $out/${tag}_tl_write/w_results.db; } > $out/${tag}_train_aspirin1024_pmc_write_size.txt
rm -rf $out/${tag}_tl_trace $out/${tag}_tl_fetch $out/${tag}_tl_write
head -30 $out/${tag}_train_aspirin1024_kernel_stats.txt
