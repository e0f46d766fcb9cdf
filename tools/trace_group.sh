#!/bin/bash
# per-kernel device time of one frame-set of a tracker group (svo_hip_tracker_group_track) -- GPU box, repo root:
#   tools/trace_group.sh <tag> <n_cameras>   ->  gpurun_out/<tag>_group_trace.txt
set -e -o pipefail
tag=${1:?tag}; n=${2:-8}
export TMPDIR=/tmp
root=$PWD
cd /tmp && rm -rf /tmp/grpprof
rocprofv3 --kernel-trace --output-format csv -d /tmp/grpprof -o grp -- python3 $root/tools/chain_bench.py --group-only --camera-counts $n > $root/gpurun_out/${tag}_group_prof.json 2> $root/gpurun_out/${tag}_group_prof.err
csv=$(find /tmp/grpprof -name '*kernel_trace.csv' | head -1)
python3 - "$csv" "$n" > $root/gpurun_out/${tag}_group_trace.txt <<'PY'
import csv, sys, collections
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
ends=[i for i,r in enumerate(rows) if 'trk_finish_cams_kernel' in r['Kernel_Name']]
lo=ends[-2]+1; hi=ends[-1]+1
t0=int(rows[lo]['Start_Timestamp'])
print("one frame-set of %s cameras through svo_hip_tracker_group_track, kernels in order (start us, duration us):" % sys.argv[2])
for r in rows[lo:hi]:
    print("%-70s %9.2f %8.2f" % (r['Kernel_Name'][:70], (int(r['Start_Timestamp'])-t0)/1e3, (int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3))
print("span us", (int(rows[hi-1]['End_Timestamp'])-t0)/1e3)
agg=collections.defaultdict(list)
for r in rows[ends[len(ends)//2]+1:]:
    agg[r['Kernel_Name'][:60]].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
print("mean duration per kernel over the second half of the run:")
for k,v in agg.items(): print("%-62s n=%4d mean %8.2f us" % (k, len(v), sum(v)/len(v)))
PY
cat $root/gpurun_out/${tag}_group_trace.txt
