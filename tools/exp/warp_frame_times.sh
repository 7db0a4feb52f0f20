#!/bin/bash
# runs on the GPU box: per-kernel times of the warp group on a cfg3 frame and a cfg5 grid (kernel trace), then the warp parity tests
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/warp_frame_${1:-x}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/cfg3 --output-format csv -- python3 $ROOT/bench.py --workload cfg3 --no-extras --cpu-rays 0 --no-psnr --steps 3 --warmup 1 > $OUT/cfg3.json 2>/dev/null
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/cfg5 --output-format csv -- python3 $ROOT/bench.py --workload cfg5 --no-extras --steps 5 --warmup 2 --cpu-rays 0 > $OUT/cfg5.json 2>/dev/null
cd $ROOT
echo "== cfg3"; python3 tools/kstats.py $OUT/cfg3 | head -12
echo "== cfg5"; python3 tools/kstats.py $OUT/cfg5 | head -10
grep -o '"ms_per_step": [0-9.]*' $OUT/cfg3.json $OUT/cfg5.json
