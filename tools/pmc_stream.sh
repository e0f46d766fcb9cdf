#!/bin/bash
# The counters roofline_jacobian_pass rests on: the default bench workload run through the STREAMING kernels
# (bench.py --stream-mode), one rocprofv3 --pmc pass per counter group, kernel-trace only (GPU box, repo root):
#   tools/pmc_stream.sh <tag>   ->  gpurun_out/<tag>_pmc_extra.txt, gpurun_out/<tag>_pmc.json
set -e -o pipefail
tag=${1:?tag}; shift || true
export PMC_ARGS="--stream-mode --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --profile-events 0 $*"
export PMC_KERNEL="sia_"
bash tools/pmc_pass.sh "$tag" \
  "SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_CVT SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES" \
  "GRBM_GUI_ACTIVE FETCH_SIZE" \
  "WRITE_SIZE" \
  "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" > /dev/null
python3 tools/pmc_json.py "gpurun_out/${tag}_pmc_extra.txt" "gpurun_out/${tag}_pmc.json" "rocprofv3 --kernel-trace --pmc <group> -- python3 bench.py $PMC_ARGS"
