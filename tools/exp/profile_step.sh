#!/bin/bash
# Runs ON THE GPU BOX: kernel trace of the replayed training step at F frames per GPU -> its timeline (tools/exp/step_timeline.py)
F=${1:-2}; shift
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/r05/prof_f$F
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/trace --output-format csv -- python3 $ROOT/bench.py --workload cfg4 --no-extras --steps 40 --warmup 5 --frames-per-gpu $F "$@" > $OUT/line.json 2> /dev/null
cd $ROOT
python3 tools/exp/step_timeline.py $OUT/trace > gpurun_out/r05/train_step_timeline_f$F.txt
rm -rf $OUT/trace
head -3 gpurun_out/r05/train_step_timeline_f$F.txt
