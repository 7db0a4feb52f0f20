# round 5 scratch runner (GPU box): the GPU suite, then the training step's timings in the configurations under study
mkdir -p gpurun_out/r05
python tools/exp/graph_external_event.py 2>&1 | tail -8
python -m pytest tests -m gpu -q -x 2>&1 | tail -25 | tee gpurun_out/r05/gpu_tests.txt
t() { python bench.py --workload cfg4 --no-extras --steps 40 --warmup 5 "$@" 2>/dev/null | grep '^{' | python -c "
import json,sys;d=json.loads(sys.stdin.read());print(round(d['ms_per_step'],3),'ms', d['config']['kernel_launches_timed_per_step'],'launches, peak', d.get('peak_alloc_GiB'),'GiB', {k:v for k,v in d['kernel_time_share'].items()})"; }
echo "cfg4 f16: $(t --frames-per-gpu 16)"
echo "cfg4 f2: $(t --frames-per-gpu 2)"
echo "cfg4 refine f16: $(t --frames-per-gpu 16 --refine)"
echo "cfg4 refine f2: $(t --frames-per-gpu 2 --refine)"
