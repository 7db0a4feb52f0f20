# A/B on configs[2] (same box): coarse depths from the step table (default) against the depth array of round 5
mkdir -p gpurun_out/r06
t() { env "$@" python bench.py --workload cfg3 --no-extras --steps 8 --warmup 2 2>/dev/null | grep '^{' | python -c "
import json,sys;d=json.loads(sys.stdin.read());k=d['roofline_hbm_kernels']['kernels'];print(round(d['ms_per_step'],3),'ms', {n:round(v['ms_per_step'],3) for n,v in k.items()})"; }
for rep in 1 2 3; do
  echo "steps : $(t A=1)"
  echo "array : $(t ANR_COARSE_DEPTH_ARRAY=1)"
done
