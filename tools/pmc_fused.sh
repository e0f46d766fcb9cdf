#!/bin/bash
# All the counters bench.py's roofline object rests on, for the default bench command (run on the GPU box, repo root):
#   tools/pmc_fused.sh <tag>   ->  gpurun_out/<tag>_pmc_extra.txt, gpurun_out/<tag>_pmc.json, gpurun_out/<tag>_kernel_stats.md
# One rocprofv3 --pmc pass per counter group (8 SQ slots / 4 TCC slots per pass), kernel-trace only.
set -e -o pipefail
tag=${1:?tag}; shift || true
export PMC_ARGS="--steps 2 --warmup 1 --no-cpu-baseline --no-secondary --profile-events 0 $*"
export PMC_KERNEL="sia_"
bash tools/pmc_pass.sh "$tag" \
  "SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT64" \
  "SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_INT32 SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAVES" \
  "SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SMEM" \
  "GRBM_GUI_ACTIVE FETCH_SIZE" \
  "WRITE_SIZE" \
  "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_BUBBLE_sum TCC_EA0_RDREQ_DRAM_sum" \
  "TCC_HIT_sum TCC_MISS_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" \
  "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TA_TA_BUSY_sum" \
  "SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU" > /dev/null
python3 tools/pmc_json.py "gpurun_out/${tag}_pmc_extra.txt" "gpurun_out/${tag}_pmc.json" "rocprofv3 --kernel-trace --pmc <group> -- python3 bench.py $PMC_ARGS"
