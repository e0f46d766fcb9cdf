#!/bin/bash
# rocprofv3 kernel trace of bench_c2.py (run on the GPU box, from the repo root):
#   tools/c2_trace.sh <tag> [bench_c2 args]  -> gpurun_out/<tag>_c2_kernel_stats.md
set -e -o pipefail
tag=${1:?tag}; shift || true
root=$PWD; out=$root/gpurun_out; mkdir -p "$out"
export TMPDIR=/tmp
cd /tmp
rm -rf /tmp/c2trace_$tag
rocprofv3 --kernel-trace --stats -d /tmp/c2trace_$tag -o trace -- python3 "$root/bench_c2.py" --no-cpu-baseline "$@" > "$out/${tag}_c2_under_rocprof.json" 2> "$out/${tag}_c2_rocprof.log"
db=$(find /tmp/c2trace_$tag -name '*.db' | head -1)
python3 "$root/tools/rocpd_summary.py" "$db" "$out/${tag}_c2_kernel_stats.md" > /dev/null
echo "kernel stats -> $out/${tag}_c2_kernel_stats.md"
