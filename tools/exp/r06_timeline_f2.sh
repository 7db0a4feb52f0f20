# the replayed 2-frame step's timeline at HEAD (tools/exp/step_timeline.py on a kernel trace)
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/r06/f2_trace
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT --output-format csv -- python3 $ROOT/bench.py --workload cfg4 --no-extras --steps 40 --warmup 5 --frames-per-gpu ${F:-2} > $OUT.json 2> /dev/null
cd $ROOT
python3 tools/exp/step_timeline.py $OUT > gpurun_out/r06/train_step_timeline_f${F:-2}.txt
rm -rf $OUT
