#!/bin/bash
# per-kernel device time of svo_hip_sia_run through NLLSSolver's other branches (tools/nlls_probe.py: one frame pair per call,
# 200 and 2000 patches, every configuration 35 times) -- GPU box, repo root:
#   tools/trace_nlls.sh <tag>   ->  gpurun_out/<tag>_nlls_kernel_stats.md
set -e -o pipefail
tag=${1:?tag}
export TMPDIR=/tmp
root=$PWD
cd /tmp && rm -rf /tmp/nllsprof
rocprofv3 --kernel-trace --stats -d /tmp/nllsprof -o nlls -- python3 $root/tools/nlls_probe.py > $root/gpurun_out/${tag}_nlls_prof.txt 2> $root/gpurun_out/${tag}_nlls_prof.err
db=$(find /tmp/nllsprof -name '*.db' | head -1)
python3 $root/tools/rocpd_summary.py "$db" $root/gpurun_out/${tag}_nlls_kernel_stats.md > /dev/null
cat $root/gpurun_out/${tag}_nlls_kernel_stats.md
