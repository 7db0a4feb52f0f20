# A/B of library builds on cfg3 / cfg2 frames (same box): bash tools/exp/ab_lib_cfg3.sh libA.so libB.so
t() { python bench.py --workload $1 --no-extras --steps 3 --warmup 1 --cpu-rays 0 --no-psnr 2>/dev/null | grep '^{' | python -c "
import json,sys;d=json.loads(sys.stdin.read());print(round(d['ms_per_step'],3),'ms')"; }
for rep in 1 2; do for lib in "$@"; do
  export ANIMNERF_HIP_LIB=$PWD/anim-nerf_amd/$lib
  echo "$lib: cfg3 $(t cfg3)  cfg2 $(t cfg2)"
done; done
