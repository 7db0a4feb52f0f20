# configs[2]: the fine call copies the coarse call's per-cell results (default) against searching its cells again
# (ANR_WARP_NO_PREV_CELLS=1); same box
mkdir -p gpurun_out/r06
t() { env "$@" python bench.py --workload cfg3 --no-extras --steps 8 --warmup 2 2>/dev/null | grep '^{' | python -c "
import json,sys;d=json.loads(sys.stdin.read());k=d['roofline_hbm_kernels']['kernels'];print(round(d['ms_per_step'],3),'ms', {n:round(v['ms_per_step'],3) for n,v in k.items()})"; }
for rep in 1 2; do
  echo "copied   : $(t A=1)"
  echo "searched : $(t ANR_WARP_NO_PREV_CELLS=1)"
done
