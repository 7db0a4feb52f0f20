# A/B of library builds on every workload (same box): bash tools/exp/ab_lib_all.sh libA.so libB.so ...
t4() { python bench.py --workload cfg4 --no-extras --steps 60 --warmup 5 --frames-per-gpu $1 2>/dev/null | grep '^{' | python -c "
import json,sys;d=json.loads(sys.stdin.read());print(round(d['ms_per_step'],3))"; }
t3() { python bench.py --workload $1 --no-extras --steps 3 --warmup 1 --cpu-rays 0 --no-psnr 2>/dev/null | grep '^{' | python -c "
import json,sys;d=json.loads(sys.stdin.read());print(round(d['ms_per_step'],3))"; }
for rep in 1 2; do for lib in "$@"; do
  export ANIMNERF_HIP_LIB=$PWD/anim-nerf_amd/$lib
  echo "$lib: f2 $(t4 2) f16 $(t4 16) cfg3 $(t3 cfg3) cfg5 $(t3 cfg5)"
done; done
