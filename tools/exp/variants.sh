#!/bin/bash
# runs on the GPU box: warp kernels per call for each experiment library given as argument
ROOT=${GRAFT_REPO_ROOT:-$PWD}
for v in "$@"; do
  export ANIMNERF_HIP_LIB=$ROOT/anim-nerf_amd/$v.so
  echo "== $v"; bash $ROOT/tools/exp/warp_calls.sh $v 2>/dev/null | head -5
done
