# kernel times of the frame kernels in the replayed step for each frames-per-workgroup setting (GPU box)
for F in 2 16; do for BT in 1 2 4; do
  ANR_FO_BT=$BT ANR_FV_BT=$BT bash tools/exp/profile_step.sh $F > /dev/null 2>&1
  echo "f$F BT=$BT: $(head -1 gpurun_out/r05/train_step_timeline_f$F.txt | cut -c1-40) $(grep -o '[0-9.]* us  q[0-9]  [a-z: ]*frame_[a-z]*' gpurun_out/r05/train_step_timeline_f$F.txt | awk '{printf "%s=%s ", $NF, $1}')"
done; done
