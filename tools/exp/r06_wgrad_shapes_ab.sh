# A/B (same box): the weight-gradient call's three GEMM shapes in ONE launch (default) against a launch per shape (ANR_WGRAD_SEPARATE=1)
t() { env "$@" python bench.py --workload cfg4 --no-extras --steps 60 --warmup 5 --frames-per-gpu $F 2>/dev/null | grep '^{' | python -c "
import json,sys;d=json.loads(sys.stdin.read());print(round(d['ms_per_step'],3),'ms')"; }
for rep in 1 2; do for v in one separate; do
  if [ $v = one ]; then e="A=1"; else e="ANR_WGRAD_SEPARATE=1"; fi
  echo "$v: f2 $(F=2 t $e)  f1 $(F=1 t $e)  f4 $(F=4 t $e)  f16 $(F=16 t $e)"
done; done
