# round 5 scratch runner (GPU box): the training suite, then the replayed step's timings
mkdir -p gpurun_out/r05
python -m pytest tests/test_gpu_training.py -q -x 2>&1 | tail -4
t() { python bench.py --workload cfg4 --no-extras --steps 60 --warmup 5 "$@" 2>/dev/null | grep '^{' | python -c "
import json,sys;d=json.loads(sys.stdin.read());print(round(d['ms_per_step'],3),'ms')"; }
echo "f2: $(t --frames-per-gpu 2)  again: $(t --frames-per-gpu 2)"
echo "f16: $(t --frames-per-gpu 16)  again: $(t --frames-per-gpu 16)"
echo "refine f2: $(t --frames-per-gpu 2 --refine)   refine f16: $(t --frames-per-gpu 16 --refine)"
