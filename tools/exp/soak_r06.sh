# round 6 soaks (GPU box): the 60-launch graphed step with half tiles / the one-launch weight-gradient shapes —
# replay reproducibility at 2 and 16 frames (tools/exp/race_hunt.py) and 20,000-step runs from both streams (tools/soak_train.py)
mkdir -p gpurun_out/r06
for f in 2 16; do
  echo "== race_hunt 20000 replays, $f frames"; timeout 900 python tools/exp/race_hunt.py 20000 $f 2>&1 | tail -4
done
echo "== race_hunt 20000 replays, 2 frames, entered from the default stream"; timeout 900 python tools/exp/race_hunt.py 20000 2 default 2>&1 | tail -4
for v in own default; do
  if [ $v = default ]; then export SOAK_FROM_DEFAULT_STREAM=1; else unset SOAK_FROM_DEFAULT_STREAM; fi
  SOAK_PRINT_EVERY=2000 timeout 900 python tools/soak_train.py 20000 graph > /tmp/soak_$$.txt 2>&1; rc=$?
  echo "== 20000 graphed steps entered from the $v stream: rc $rc"; grep "^step" /tmp/soak_$$.txt | awk 'NR==1 || NR%3==0' | tail -6; tail -2 /tmp/soak_$$.txt
done
