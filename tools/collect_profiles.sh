#!/bin/bash
# Collect the evidence bench.py's roofline object rests on (run on the GPU box, from the repo root):
#   1. rocprofv3 --kernel-trace --stats of the default bench command -> gpurun_out/<tag>_kernel_stats.md
#   2. two separate --pmc passes (FETCH_SIZE, WRITE_SIZE) of a short run -> gpurun_out/<tag>_pmc.txt
#   3. the bench JSON line itself                                    -> gpurun_out/<tag>_bench.json
# Usage: tools/collect_profiles.sh <tag> [extra bench.py args]
# The summaries to be judged are then copied by hand into profiles/ (gpurun_out/ is scratch).
set -e -o pipefail
tag=${1:?tag}
shift || true
root=$PWD
out=$root/gpurun_out
mkdir -p "$out"
export TMPDIR=/tmp
cd /tmp
rm -rf /tmp/prof_$tag
# 1. kernel trace (program directly after --, no launcher in between)
rocprofv3 --kernel-trace --stats -d /tmp/prof_$tag/trace -o trace -- python3 "$root/bench.py" --steps 20 --warmup 3 --no-cpu-baseline --no-secondary "$@" > "$out/${tag}_bench_under_rocprof.json" 2> "$out/${tag}_rocprof.log"
db=$(find /tmp/prof_$tag/trace -name '*.db' | head -1)
python3 "$root/tools/rocpd_summary.py" "$db" "$out/${tag}_kernel_stats.md" > /dev/null
echo "kernel stats -> $out/${tag}_kernel_stats.md"
# 2. PMC passes, each in its own run, with kernel-trace only
: > "$out/${tag}_pmc.txt"
for ctr in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d /tmp/prof_$tag/$ctr -o pmc -- python3 "$root/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --profile-events 0 "$@" > /dev/null 2>> "$out/${tag}_rocprof.log"
  csv=$(find /tmp/prof_$tag/$ctr -name '*counter_collection.csv' | head -1)
  python3 "$root/tools/pmc_summary.py" "$csv" sia_ >> "$out/${tag}_pmc.txt"
done
echo "pmc -> $out/${tag}_pmc.txt"
# 3. the plain bench line (no profiler attached)
cd "$root"
python3 bench.py "$@" > "$out/${tag}_bench.json" 2>> "$out/${tag}_rocprof.log"
cat "$out/${tag}_bench.json"
