# PMC passes over tools/bench_composite_sample.py (GPU box): where does composite_sample_kernel spend its cycles?
set -u
ROOT=${GRAFT_REPO_ROOT:-$PWD}; OUT=$ROOT/gpurun_out/pmc_cs; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT" "SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_LDS_IDX_ACTIVE"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  timeout 300 rocprofv3 --pmc $set --kernel-trace -d $OUT/$tag --output-format csv -- python3 $ROOT/tools/bench_composite_sample.py 3 > /dev/null 2>&1
  python3 $ROOT/tools/pmc_summary.py $OUT/$tag composite_sample_kernel
done
