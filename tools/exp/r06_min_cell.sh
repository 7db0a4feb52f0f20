# sensitivity of the cell-sorted search to the cell size: MIN_CELL 30 mm (product since this A/B: the bench's body + 2 x dis_threshold
# over 64 cells = 3.2-cm cells) against 40 mm (the variant: build.build(defines=['ANR_MIN_CELL_MM=40'], out='.../libanimnerf_hip.mincell40.so')); kernel trace of a configs[2] frame, the eight largest calls per kernel, us; same box
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r06
cd /tmp && export TMPDIR=/tmp
for v in 30 40 30 40; do
  if [ $v = 40 ]; then export ANIMNERF_HIP_LIB=$R/anim-nerf_amd/libanimnerf_hip.mincell40.so; else unset ANIMNERF_HIP_LIB; fi
  rm -rf $R/gpurun_out/r06/mc_$v
  rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r06/mc_$v --output-format csv -- python3 $R/bench.py --workload cfg3 --no-extras --steps 3 --warmup 1 > $R/gpurun_out/r06/mc_$v.json 2>/dev/null
  f=$(find $R/gpurun_out/r06/mc_$v -name '*kernel_trace.csv' | head -1)
  python3 - "$f" $v $R/gpurun_out/r06/mc_$v.json <<'P'
import csv,sys,collections,json
by=collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    by[r['Kernel_Name'][:44]].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
d=json.loads([l for l in open(sys.argv[3]) if l.startswith('{')][0])
print('cell', sys.argv[2], 'mm: frame', round(d['ms_per_step'],3), 'ms, warp_points', round(d['roofline_hbm_kernels']['kernels']['warp_points']['ms_per_step'],3))
for k,v in by.items():
    if any(t in k for t in ('warp_search_kernel','warp_cells_kernel','classify_lean_kernel<true>','scatter')): print('  ', k, sorted(round(x,1) for x in v)[-8:])
P
  rm -rf $R/gpurun_out/r06/mc_$v $R/gpurun_out/r06/mc_$v.json
done
