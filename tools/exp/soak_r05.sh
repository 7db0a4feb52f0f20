# round 5 soaks of the 70-launch graphed step (GPU box): from the Trainer's stream and from torch's default stream
mkdir -p gpurun_out/r05
for v in own default own default; do
  if [ $v = default ]; then export SOAK_FROM_DEFAULT_STREAM=1; else unset SOAK_FROM_DEFAULT_STREAM; fi
  SOAK_PRINT_EVERY=2000 timeout 900 python tools/soak_train.py 20000 graph > /tmp/soak_$$.txt 2>&1; rc=$?
  echo "== 20000 graphed steps entered from the $v stream: rc $rc"; grep "^step" /tmp/soak_$$.txt | awk 'NR==1 || NR%3==0' | tail -6; tail -2 /tmp/soak_$$.txt
done
