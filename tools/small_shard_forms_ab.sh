#!/usr/bin/env bash
# Small shards: the edge MLPs in their row-local form (default below ~830 pair tiles) against the persistent / register-weights
# forms forced at every size (NNHIP_MLP_WIDE_TILES=0), one stream and two steps in flight.  usage (through gpurun): tools/small_shard_forms_ab.sh
for tiles in default 0 200; do
  if [ $tiles = default ]; then unset NNHIP_MLP_WIDE_TILES; else export NNHIP_MLP_WIDE_TILES=$tiles; fi
  echo "## NNHIP_MLP_WIDE_TILES=$tiles"
  python tools/two_stream_ab.py 64 128 256 512 2>&1 | grep "^B ="
done
