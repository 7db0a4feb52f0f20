# the warp passes of a configs[2] frame by the kernel trace (3 + 1 frames): us per call, the eight largest calls of each kernel
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r06
cd /tmp && export TMPDIR=/tmp
for v in a b; do
  rm -rf $R/gpurun_out/r06/wt_$v
  rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r06/wt_$v --output-format csv -- python3 $R/bench.py --workload cfg3 --no-extras --steps 3 --warmup 1 > /dev/null 2>&1
  f=$(find $R/gpurun_out/r06/wt_$v -name '*kernel_trace.csv' | head -1)
  python3 - "$f" $v <<'P'
import csv,sys,collections
by=collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    by[r['Kernel_Name'][:52]].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
for k,v in by.items():
    if 'anr::warp' in k: print(sys.argv[2], k, len(v), 'largest 8:', sorted(round(x,1) for x in v)[-8:])
P
  rm -rf $R/gpurun_out/r06/wt_$v
done
