#!/bin/bash
# Runs ON THE GPU BOX: the configs[2] part of tools/collect_profiles.sh only (kernel trace + FETCH_SIZE / WRITE_SIZE passes)
set -u
ROOT=${GRAFT_REPO_ROOT:-$PWD}; OUT=$ROOT/gpurun_out/prof_cfg3_head; DST=$ROOT/gpurun_out/profiles_cfg3_head; mkdir -p $OUT $DST
cd /tmp && export TMPDIR=/tmp
extra="--no-extras --cpu-rays 0 --no-psnr --steps 3 --warmup 1"
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/cfg3_trace --output-format csv -- python3 $ROOT/bench.py --workload cfg3 $extra > $OUT/cfg3_trace.json 2> /dev/null
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/cfg3_fetch --output-format csv -- python3 $ROOT/bench.py --workload cfg3 $extra > /dev/null 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/cfg3_write --output-format csv -- python3 $ROOT/bench.py --workload cfg3 $extra > /dev/null 2>&1
cd $ROOT
{ echo "== cfg3 (4 frames: 1 warm-up + 3 timed)"; python3 tools/hbm_table.py $OUT/cfg3_fetch $OUT/cfg3_write $OUT/cfg3_trace; } > $DST/hbm_cfg3.txt
cp $OUT/cfg3_trace/*/*kernel_stats.csv $DST/bench_cfg3_bf16_kernel_stats.csv
grep '^{' $OUT/cfg3_trace.json > $DST/bench_cfg3_bf16_under_rocprof.json
head -16 $DST/hbm_cfg3.txt
