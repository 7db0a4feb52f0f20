#!/bin/bash
# runs on the GPU box: issue counters of the warp kernels on a cfg3 frame (one --pmc pass, kernel trace only)
ROOT=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
OUT=$ROOT/gpurun_out/warp_pmc_${1:-x}; mkdir -p $OUT
timeout 600 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES --kernel-trace -d $OUT/p1 --output-format csv -- python3 $ROOT/bench.py --workload cfg3 --no-extras --steps 1 --warmup 1 --cpu-rays 0 --no-psnr > /dev/null 2>&1
timeout 600 rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA --kernel-trace -d $OUT/p2 --output-format csv -- python3 $ROOT/bench.py --workload cfg3 --no-extras --steps 1 --warmup 1 --cpu-rays 0 --no-psnr > /dev/null 2>&1
cd $ROOT
python3 tools/pmc_summary.py $OUT/p1 warp_search_kernel warp_classify warp_cells
python3 tools/pmc_summary.py $OUT/p2 warp_search_kernel warp_classify warp_cells
