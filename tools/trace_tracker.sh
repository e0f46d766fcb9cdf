#!/bin/bash
# per-kernel device time of one tracked frame (svo_hip_tracker_track) -- run on the GPU box from the repo root:
#   tools/trace_tracker.sh <tag> [--min-level 2]  ->  gpurun_out/<tag>_tracker_trace.txt
set -e -o pipefail
tag=${1:?tag}; shift || true
export TMPDIR=/tmp
root=$PWD
cd /tmp && rm -rf /tmp/trkprof
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/trkprof -o trk -- python3 $root/tools/chain_bench.py --tracker-only "$@" > $root/gpurun_out/${tag}_tracker_prof.json 2> $root/gpurun_out/${tag}_tracker_prof.err
csv=$(find /tmp/trkprof -name '*kernel_trace.csv' | head -1)
python3 - "$csv" > $root/gpurun_out/${tag}_tracker_trace.txt <<'PY'
import csv, sys, collections
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
# the last 19 frames: find the last occurrences of trk_finish_kernel
ends=[i for i,r in enumerate(rows) if 'trk_finish_kernel' in r['Kernel_Name']]
lo=ends[-2]+1; hi=ends[-1]+1
t0=int(rows[lo]['Start_Timestamp'])
print("one frame, kernels in order (start us, duration us):")
for r in rows[lo:hi]:
    print("%-70s %9.2f %8.2f" % (r['Kernel_Name'][:70], (int(r['Start_Timestamp'])-t0)/1e3, (int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3))
print("span us", (int(rows[hi-1]['End_Timestamp'])-t0)/1e3)
agg=collections.defaultdict(list)
for r in rows[ends[len(ends)//2]+1:]:
    agg[r['Kernel_Name'][:60]].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
print("mean duration per kernel over the second half of the run:")
for k,v in agg.items(): print("%-62s n=%4d mean %8.2f us" % (k, len(v), sum(v)/len(v)))
PY
cat $root/gpurun_out/${tag}_tracker_trace.txt
