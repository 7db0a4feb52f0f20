# A/B on configs[1] (same box): the staged launches (default) against the one-pass ray-march kernel (ANR_ONE_PASS=1)
t() { env "$@" python bench.py --no-extras --steps 6 --warmup 2 2>/dev/null | grep '^{' | python -c "
import json,sys;d=json.loads(sys.stdin.read());print(round(d['ms_per_step'],3),'ms', round(d['value']/1e6,3),'M rays/s')"; }
for rep in 1 2; do
  echo "staged  : $(t A=1)"
  echo "one pass: $(t ANR_ONE_PASS=1)"
done
