#!/bin/bash
# GPU box: average time of the kernels matching $1 in a configs[2] frame, per library build: kernel_time_cfg3.sh PATTERN lib...
PAT=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$PWD}; cd /tmp && export TMPDIR=/tmp
for lib in "$@"; do
  export ANIMNERF_HIP_LIB=$ROOT/anim-nerf_amd/$lib
  rm -rf /tmp/kt_$$; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/kt_$$ --output-format csv -- python3 $ROOT/bench.py --workload cfg3 --no-extras --cpu-rays 0 --no-psnr --steps 3 --warmup 1 > /dev/null 2>&1
  echo "$lib: $(python3 $ROOT/tools/kstats.py /tmp/kt_$$ 4 40 | grep "$PAT" | cut -c1-60,70-)"
done
