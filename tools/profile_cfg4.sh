#!/bin/bash
# ON THE GPU BOX: kernel trace of the graphed cfg4 training step at F frames per GPU -> per-step launches, idle time, kernel ranking
F=${1:-16}; OUT=gpurun_out/cfg4_f${F}_trace
cd /tmp && export TMPDIR=/tmp && cd ${GRAFT_REPO_ROOT:-/root/repo}
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats -d $OUT --output-format csv -- python3 bench.py --workload cfg4 --no-extras --steps 40 --warmup 5 --frames-per-gpu $F > $OUT/bench.json 2> $OUT/err.txt
python3 tools/step_gaps.py $OUT > gpurun_out/cfg4_f${F}_step_gaps.txt 2>&1
cat gpurun_out/cfg4_f${F}_step_gaps.txt
