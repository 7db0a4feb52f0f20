# round 5 scratch runner (GPU box): the replayed step under different branch structures
mkdir -p gpurun_out/r05
t() { python bench.py --workload cfg4 --no-extras --steps 60 --warmup 5 "$@" 2>/dev/null | grep '^{' | python -c "
import json,sys;d=json.loads(sys.stdin.read());print(round(d['ms_per_step'],3),'ms')"; }
for f in 2 16; do
echo "f$f default: $(t --frames-per-gpu $f)  again: $(t --frames-per-gpu $f)"
echo "f$f wgrad fork first: $(ANR_STEP_WGRAD_FIRST=1 t --frames-per-gpu $f)"
echo "f$f one side stream: $(ANR_STEP_ONE_SIDE=1 t --frames-per-gpu $f)"
echo "f$f one side + wgrad first: $(ANR_STEP_ONE_SIDE=1 ANR_STEP_WGRAD_FIRST=1 t --frames-per-gpu $f)"
echo "f$f no branches: $(ANR_STEP_BRANCHES=0 t --frames-per-gpu $f)"
done
python -m pytest tests/test_gpu_training.py -q 2>&1 | tail -8
