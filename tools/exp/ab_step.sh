# A/B of env settings on the replayed cfg4 step (same box): usage  bash tools/exp/ab_step.sh "K=V" ["K2=V2" ...]
mkdir -p gpurun_out/r05
t() { env "$@" python bench.py --workload cfg4 --no-extras --steps 60 --warmup 5 --frames-per-gpu $F 2>/dev/null | grep '^{' | python -c "
import json,sys;d=json.loads(sys.stdin.read());print(round(d['ms_per_step'],3),'ms')"; }
for F in 2 16; do
  for rep in 1 2; do
    echo "f$F base: $(t A=1)"
    for kv in "$@"; do echo "f$F $kv: $(t $kv)"; done
  done
done
