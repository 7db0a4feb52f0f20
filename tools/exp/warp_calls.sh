#!/bin/bash
# runs on the GPU box: warp kernels per call on a cfg3 frame and a cfg5 grid
ROOT=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
OUT=$ROOT/gpurun_out/deal_${1:-x}; mkdir -p $OUT
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/cfg3 --output-format csv -- python3 $ROOT/bench.py --workload cfg3 --no-extras --cpu-rays 0 --no-psnr --steps 3 --warmup 1 > $OUT/cfg3.json 2>/dev/null
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/cfg5 --output-format csv -- python3 $ROOT/bench.py --workload cfg5 --no-extras --steps 5 --warmup 2 --cpu-rays 0 > $OUT/cfg5.json 2>/dev/null
python3 $ROOT/tools/exp/search_calls.py $OUT/cfg3; python3 $ROOT/tools/kstats.py $OUT/cfg5 | grep "warp_search\|warp_cells"
grep -o '"ms_per_step": [0-9.]*' $OUT/cfg3.json | head -1; grep -o '"ms_per_step": [0-9.]*' $OUT/cfg5.json | head -1
