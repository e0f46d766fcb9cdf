#!/bin/bash
# A/B of counters between kernel builds on the same box: tools/pmc_ab.sh "<counter group>" libtag1 libtag2 ... (build/libsvo_hip_<tag>.so)
set -e -o pipefail
grp=$1; shift
export PMC_ARGS="--steps 2 --warmup 1 --no-cpu-baseline --no-secondary --profile-events 0"
export PMC_KERNEL="sia_fused"
for lib in "$@"; do
  SVO_HIP_LIB=$PWD/build/libsvo_hip_$lib.so bash tools/pmc_pass.sh ab_$lib "$grp" > /dev/null
  echo "== $lib"; cat gpurun_out/ab_${lib}_pmc_extra.txt
done
