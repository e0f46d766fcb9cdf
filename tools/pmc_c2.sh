#!/bin/bash
# PMC passes over the depth-filter kernels (df_geometry / df_search / df_align / df_finalize) at BASELINE config C2's size
# (100 000 seeds, 640x480) and at config C4's on one GPU (1 000 000 seeds, 1280x720) -- run on the GPU box, repo root:
#   tools/pmc_c2.sh <tag>  ->  gpurun_out/<tag>_pmc_df_c2.json, gpurun_out/<tag>_pmc_df_c4.json (+ .txt)
# Same counter groups as tools/pmc_fused.sh; one rocprofv3 --kernel-trace --pmc pass per group, nothing else traced.
set -e -o pipefail
tag=${1:?tag}
export PMC_SCRIPT=tools/dfbench_only.py
export PMC_KERNEL="df_"
groups=(
  "SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT64"
  "SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_INT32 SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAVES"
  "SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SMEM"
  "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM SQ_INSTS_FLAT"
  "GRBM_GUI_ACTIVE FETCH_SIZE"
  "WRITE_SIZE"
  "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum"
  "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TA_TA_BUSY_sum TA_BUSY_avr"
  # lane utilisation of the VALU (rocprof's VALUUtilization = SQ_THREAD_CYCLES_VALU / (64 x SQ_ACTIVE_INST_VALU)): the issue
  # fractions above count a masked lane as busy; both counters from ONE pass, last, so that this pair is what the JSON keeps
  "SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU"
)
PMC_ARGS="100000 640 480 0 4" bash tools/pmc_pass.sh "${tag}_df_c2" "${groups[@]}" > /dev/null
python3 tools/pmc_json.py "gpurun_out/${tag}_df_c2_pmc_extra.txt" "gpurun_out/${tag}_pmc_df_c2.json" \
  "rocprofv3 --kernel-trace --pmc <group> -- python3 tools/dfbench_only.py 100000 640 480 0 4" df
PMC_ARGS="1000000 1280 720 0.0045 4" bash tools/pmc_pass.sh "${tag}_df_c4" "${groups[@]}" > /dev/null
python3 tools/pmc_json.py "gpurun_out/${tag}_df_c4_pmc_extra.txt" "gpurun_out/${tag}_pmc_df_c4.json" \
  "rocprofv3 --kernel-trace --pmc <group> -- python3 tools/dfbench_only.py 1000000 1280 720 0.0045 4" df
