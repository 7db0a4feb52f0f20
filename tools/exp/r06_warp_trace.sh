# the warp passes of a configs[2] frame by the kernel trace (3 + 1 frames): us per call, the eight largest calls of each kernel
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r06
cd /tmp && export TMPDIR=/tmp
for v in a b; do
  rm -rf $R/gpurun_out/r06/wt_$v
  rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r06/wt_$v --output-format csv -- python3 $R/bench.py --workload cfg3 --no-extras --steps 3 --warmup 1 > $R/gpurun_out/r06/wt_$v.json 2>/dev/null
  f=$(find $R/gpurun_out/r06/wt_$v -name '*kernel_trace.csv' | head -1)
  python3 - "$f" $v $R/gpurun_out/r06/wt_$v.json <<'P'
import csv,sys,collections,json
by=collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    by[r['Kernel_Name'][:52]].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
d=json.loads([l for l in open(sys.argv[3]) if l.startswith('{')][0])
print(sys.argv[2], ': frame', round(d['ms_per_step'],3), 'ms, warp_points', round(d['roofline_hbm_kernels']['kernels']['warp_points']['ms_per_step'],3))
for k,v in by.items():
    if 'anr::warp' in k: print('  ', k, len(v), 'largest 8:', sorted(round(x,1) for x in v)[-8:])
P
  rm -rf $R/gpurun_out/r06/wt_$v $R/gpurun_out/r06/wt_$v.json
done
