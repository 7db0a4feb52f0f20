# PMC passes over a configs[2] frame (GPU box): instruction and wait counters of the warp kernels
set -u
ROOT=${GRAFT_REPO_ROOT:-$PWD}; OUT=$ROOT/gpurun_out/pmc_cfg3; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  timeout 300 rocprofv3 --pmc $set --kernel-trace -d $OUT/$tag --output-format csv -- python3 $ROOT/bench.py --workload cfg3 --no-extras --cpu-rays 0 --no-psnr --steps 3 --warmup 1 > /dev/null 2>&1
  python3 $ROOT/tools/pmc_summary.py $OUT/$tag warp_search_kernel warp_classify_lean warp_cells_kernel
done
