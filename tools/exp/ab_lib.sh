# A/B of library builds on the replayed cfg4 step and the small-batch warp (same box): bash tools/exp/ab_lib.sh libA.so libB.so ...
t() { python bench.py --workload cfg4 --no-extras --steps 60 --warmup 5 --frames-per-gpu $F 2>/dev/null | grep '^{' | python -c "
import json,sys;d=json.loads(sys.stdin.read());print(round(d['ms_per_step'],3),'ms')"; }
for rep in 1 2; do for lib in "$@"; do
  export ANIMNERF_HIP_LIB=$PWD/anim-nerf_amd/$lib
  echo "$lib: warp2 $(python tools/bench_warp_small.py 30 groups 2 2>/dev/null | tail -1 | cut -c1-30)  warp16 $(python tools/bench_warp_small.py 30 groups 16 2>/dev/null | tail -1 | cut -c1-30)  f2 $(F=2 t)  f16 $(F=16 t)"
done; done
