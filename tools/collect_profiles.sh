#!/bin/bash
# Runs ON THE GPU BOX (gpurun -- bash tools/collect_profiles.sh): kernel-trace + PMC passes of the bench command for the
# round's profiles.  rocprofv3 gets `python3 bench.py ...` directly (no wrapper between the profiler and the program);
# --pmc passes are separate from each other and carry no trace domain besides --kernel-trace.
set -u
ROOT=${GRAFT_REPO_ROOT:-$PWD}
R=${ANR_ROUND:-r06}
OUT=$ROOT/gpurun_out/prof_$R
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for wl in cfg2 cfg3 cfg4; do
  extra="--no-extras --cpu-rays 0 --no-psnr --steps 3 --warmup 1"
  [ $wl = cfg4 ] && extra="--no-extras --steps 30 --warmup 2"
  timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/${wl}_trace --output-format csv -- python3 $ROOT/bench.py --workload $wl $extra > $OUT/${wl}_trace.json 2> /dev/null
  if [ $wl != cfg4 ]; then
    timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/${wl}_fetch --output-format csv -- python3 $ROOT/bench.py --workload $wl $extra > /dev/null 2>&1
    timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/${wl}_write --output-format csv -- python3 $ROOT/bench.py --workload $wl $extra > /dev/null 2>&1
  fi
done
cd $ROOT
for wl in cfg2 cfg3; do
  echo "== $wl (4 frames: 1 warm-up + 3 timed)" > $OUT/hbm_${wl}.txt
  python3 tools/hbm_table.py $OUT/${wl}_fetch $OUT/${wl}_write $OUT/${wl}_trace >> $OUT/hbm_${wl}.txt
  cat $OUT/hbm_${wl}.txt
done
# the files the judge reads, named as in profiles/r03/ .. r05/
DST=$ROOT/gpurun_out/profiles_$R
mkdir -p $DST
for wl in cfg2 cfg3 cfg4; do
  cp $OUT/${wl}_trace/*/*kernel_stats.csv $DST/bench_${wl}_bf16_kernel_stats.csv
  grep '^{' $OUT/${wl}_trace.json > $DST/bench_${wl}_bf16_under_rocprof.json
done
cp $OUT/hbm_cfg2.txt $OUT/hbm_cfg3.txt $DST/
python3 tools/traffic_json.py $OUT ${ANR_COMMIT:-unknown} 4 > $DST/mlp_hbm_traffic.json
python3 tools/time_train_cfg4.py 2>/dev/null | tail -3 > $DST/train_step_launches.txt
python3 bench.py 2>/dev/null | grep '^{' > $DST/bench_default_line.json
ls $DST
# round 3 additions: the training MLP kernels by row count, the MLP kernel's matrix-pipe occupancy at HEAD
python3 tools/bench_train_kernels.py > $DST/train_kernels_scaling.txt 2>/dev/null
cd /tmp
timeout 600 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --kernel-trace -d $OUT/mlp_pmc --output-format csv -- python3 $ROOT/tools/bench_mlp.py 4194304 3 bf16 > $OUT/mlp_pmc.txt 2>&1
cd $ROOT
python3 tools/pmc_summary.py $OUT/mlp_pmc mlp_kernel > $DST/mlp_pmc_head.txt
python3 tools/bench_mlp.py 4194304 7 bf16,bf16_w4,f32 >> $DST/mlp_pmc_head.txt
# round 5: the 8-wave x 32-point and the 4-wave x 64-point shapes A/B on THIS box, alternating (box spread exceeds the recorded deltas)
{ echo "same-box A/B, alternating runs (tools/bench_mlp.py 4194304 7 <mode>):"; for i in 1 2 3; do python3 tools/bench_mlp.py 4194304 7 bf16,bf16_w4 2>/dev/null | tail -2; done; } > $DST/mlp_shape_ab.txt
ls $DST
# second half of round 3: the training step as one graph replay (idle time between its kernels), the small-batch warp search
python3 tools/step_gaps.py $OUT/cfg4_trace > $DST/train_step_graph_gaps.txt
python3 tools/bench_warp_small.py 20 both 2>/dev/null | tail -2 > $DST/warp_small_batch.txt
ls $DST
# round 4: the step of the reference's 8-GPU partitioning (2 frames per rank), the sigma grid per kernel, the ordered launch
# list of one replayed step, per-test durations of the GPU suite, the small-batch search at 2 and 16 bodies
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/cfg4_f2_trace --output-format csv -- python3 $ROOT/bench.py --workload cfg4 --no-extras --steps 40 --warmup 5 --frames-per-gpu 2 > $OUT/cfg4_f2_trace.json 2> /dev/null
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/cfg5_trace --output-format csv -- python3 $ROOT/bench.py --workload cfg5 --no-extras --steps 5 --warmup 2 --cpu-rays 0 > $OUT/cfg5_trace.json 2> /dev/null
cd $ROOT
python3 tools/step_gaps.py $OUT/cfg4_f2_trace > $DST/train_step_graph_gaps_f2.txt
python3 tools/step_sequence.py $OUT/cfg4_trace > $DST/train_step_launch_sequence.txt
python3 tools/kstats.py $OUT/cfg5_trace > $DST/cfg5_kernel_stats.txt 2>/dev/null
grep '^{' $OUT/cfg5_trace.json > $DST/bench_cfg5_bf16_under_rocprof.json
grep '^{' $OUT/cfg4_f2_trace.json > $DST/bench_cfg4_f2_bf16_under_rocprof.json
for b in 2 16; do echo "bodies $b: $(python3 tools/bench_warp_small.py 30 groups $b 2>/dev/null | tail -1)"; done > $DST/warp_small_batch_by_bodies.txt
python3 tools/step_ops.py 16 2>/dev/null | tail -4 > $DST/train_step_framework_ops.txt
python3 -m pytest tests -m gpu -q --durations=15 2>&1 | tail -22 > $DST/gpu_test_durations.txt
ls $DST
# round 4, second half: the replayed step's parallel branches (which launches run next to which)
python3 tools/exp/step_timeline.py $OUT/cfg4_trace > $DST/train_step_timeline.txt
python3 tools/exp/step_timeline.py $OUT/cfg4_f2_trace > $DST/train_step_timeline_f2.txt

# round 5: the frozen-network (_refine) step, the training kernels per row count, the replayed step at HEAD
python3 bench.py --workload cfg4 --no-extras --steps 40 --warmup 5 --refine 2>/dev/null | grep '^{' > $DST/bench_cfg4_refine_bf16.json
python3 bench.py --workload cfg4 --no-extras --steps 40 --warmup 5 --refine --frames-per-gpu 2 2>/dev/null | grep '^{' > $DST/bench_cfg4_refine_f2_bf16.json
# (the experiment outputs of the round — gpurun_out/r05/*.txt — are copied into profiles/r05/ on the build host)
ls $DST

# round 6: the one-pass ray-march kernel under the kernel trace (ONE launch per frame) and next to the staged launches on this
# box; the training MLP passes at small row counts (half tiles); the 2-frame step's timeline comes from the cfg4_f2 trace above
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/cfg2_one_pass_trace --output-format csv -- python3 $ROOT/bench.py --one-pass --no-extras --cpu-rays 0 --no-psnr --steps 3 --warmup 1 > $OUT/cfg2_one_pass_trace.json 2> /dev/null
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/cfg2_one_pass_fetch --output-format csv -- python3 $ROOT/bench.py --one-pass --no-extras --cpu-rays 0 --no-psnr --steps 3 --warmup 1 > /dev/null 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/cfg2_one_pass_write --output-format csv -- python3 $ROOT/bench.py --one-pass --no-extras --cpu-rays 0 --no-psnr --steps 3 --warmup 1 > /dev/null 2>&1
cd $ROOT
cp $OUT/cfg2_one_pass_trace/*/*kernel_stats.csv $DST/bench_cfg2_one_pass_bf16_kernel_stats.csv
grep '^{' $OUT/cfg2_one_pass_trace.json > $DST/bench_cfg2_one_pass_bf16_under_rocprof.json
{ echo "== cfg2 through the one-pass kernel (4 frames: 1 warm-up + 3 timed)"; python3 tools/hbm_table.py $OUT/cfg2_one_pass_fetch $OUT/cfg2_one_pass_write $OUT/cfg2_one_pass_trace; } > $DST/hbm_cfg2_one_pass.txt 2>&1
bash tools/exp/r06_one_pass_ab.sh > $DST/ab_one_pass_collection_box.txt 2>&1
python3 tools/bench_mlp_small.py > $DST/mlp_small_half_tiles_collection_box.txt 2>/dev/null
ls $DST
