#!/bin/bash
# ON THE GPU BOX: kernel-trace of a cfg3 frame with each classify ablation library -> classify / search time per call
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
for v in product ANR_ABL_CLS_NOHASH ANR_ABL_CLS_NOPTS ANR_ABL_CLS_NOREUSE ANR_ABL_CLS_NOLIST; do
  if [ $v = product ]; then unset ANIMNERF_HIP_LIB; else export ANIMNERF_HIP_LIB=$R/build/lib_$v.so; fi
  rm -rf /tmp/tr_$v
  rocprofv3 --kernel-trace --stats -d /tmp/tr_$v --output-format csv -- python3 $R/bench.py --workload cfg3 --no-extras --cpu-rays 0 --no-psnr --steps 3 --warmup 1 > /dev/null 2>&1
  echo "== $v"
  python3 $R/tools/kstats.py /tmp/tr_$v 2>/dev/null | grep -i "classify\|warp_search\|warp_cell\|valid_list" | cut -c1-150
done
