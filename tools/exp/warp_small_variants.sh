#!/bin/bash
# ON THE GPU BOX: the small-batch neighbour search (warp_search_groups_kernel) at 2 and 16 bodies for a few (cursors per body,
# list entries per trip) settings — is its ~0.3 ms at 2 bodies work, or trips to the cursors?
cd ${GRAFT_REPO_ROOT:-$PWD}
for v in "8 8" "16 8" "16 4" "8 16" "16 16" "4 8"; do
  set -- $v
  lib=anim-nerf_amd/libanimnerf_hip.seg$1_pool$2.so
  if [ ! -f $lib ]; then python anim-nerf_amd/build.py -DANR_GSEG=$1 -DANR_GPOOL=$2 --out=$lib > /dev/null 2>&1; fi
  for b in 2 16; do
    echo "cursors $1 pool $2 bodies $b: $(ANIMNERF_HIP_LIB=$PWD/$lib python tools/bench_warp_small.py 30 groups $b 2>/dev/null | tail -1)"
  done
done
