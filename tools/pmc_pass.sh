#!/bin/bash
# One rocprofv3 --pmc pass per counter group of a short bench run (run on the GPU box, from the repo root):
#   tools/pmc_pass.sh <tag> "<CTR1 CTR2>" "<CTR3>" ... -> gpurun_out/<tag>_pmc_extra.txt
# PMC_SCRIPT / PMC_ARGS / PMC_KERNEL select another program than bench.py and another kernel-name filter.
set -e -o pipefail
tag=${1:?tag}; shift
root=$PWD; out=$root/gpurun_out; mkdir -p "$out"
export TMPDIR=/tmp
cd /tmp
: > "$out/${tag}_pmc_extra.txt"
i=0
for grp in "$@"; do
  i=$((i+1))
  rm -rf /tmp/pmcx_${tag}_$i
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d /tmp/pmcx_${tag}_$i -o pmc -- python3 "$root/${PMC_SCRIPT:-bench.py}" ${PMC_ARGS---steps 2 --warmup 1 --no-cpu-baseline --profile-events 0} > /dev/null 2>> "$out/${tag}_pmc_extra.log" || { echo "group '$grp' failed" >> "$out/${tag}_pmc_extra.txt"; continue; }
  csv=$(find /tmp/pmcx_${tag}_$i -name '*counter_collection.csv' | head -1)
  python3 "$root/tools/pmc_summary.py" "$csv" "${PMC_KERNEL-sia_fused}" >> "$out/${tag}_pmc_extra.txt"
done
cat "$out/${tag}_pmc_extra.txt"
