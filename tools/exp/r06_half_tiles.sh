# A/B of the half tiles (same box).  The library without them is built first, on the build host:
#   python anim-nerf_amd/build.py -DANR_HALF_TILES=0 --out=$PWD/anim-nerf_amd/libanimnerf_hip.nohalf.so
mkdir -p gpurun_out/r06
{
echo "== half tiles"; python tools/bench_mlp_small.py
echo "== no half tiles"; ANIMNERF_HIP_LIB=$PWD/anim-nerf_amd/libanimnerf_hip.nohalf.so python tools/bench_mlp_small.py
echo "== f32 half"; python tools/bench_mlp_small.py f32 | head -4
} > gpurun_out/r06/mlp_small.txt 2>&1
t() { python bench.py --workload cfg4 --no-extras --steps 60 --warmup 5 --frames-per-gpu $F 2>/dev/null | grep '^{' | python -c "
import json,sys;d=json.loads(sys.stdin.read());print(round(d['ms_per_step'],3),'ms')"; }
{
for rep in 1 2; do for lib in libanimnerf_hip.so libanimnerf_hip.nohalf.so; do
  export ANIMNERF_HIP_LIB=$PWD/anim-nerf_amd/$lib
  echo "$lib: f2 $(F=2 t)  f1 $(F=1 t) f4 $(F=4 t) f16 $(F=16 t)"
done; done
unset ANIMNERF_HIP_LIB
} > gpurun_out/r06/ab_half_tiles.txt 2>&1
python -m pytest tests/test_gpu_training.py -x -q 2>&1 | tail -5 > gpurun_out/r06/pytest_train.txt
python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -5 > gpurun_out/r06/pytest_parity.txt
python bench.py --workload cfg3 --no-extras --steps 6 --warmup 2 > gpurun_out/r06/cfg3_steps.json 2>/dev/null
python - <<'P'
import json
d=json.load(open('gpurun_out/r06/cfg3_steps.json'))
print('cfg3', d['ms_per_step'], d.get('kernel_time_share'), d.get('roofline_hbm_kernels',{}).get('kernels'))
P
