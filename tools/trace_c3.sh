#!/bin/bash
# per-kernel device time of the patch-sharded solve on one rank (BASELINE config C3's shape) -- GPU box, repo root:
#   tools/trace_c3.sh <tag>  ->  gpurun_out/<tag>_c3_trace.txt
set -e -o pipefail
tag=${1:?tag}; shift || true
export TMPDIR=/tmp
root=$PWD
cd /tmp && rm -rf /tmp/c3prof
RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 rocprofv3 --kernel-trace --output-format csv -d /tmp/c3prof -o c3 -- python3 $root/bench.py --gpus 1 --mode allreduce --width 1280 --height 720 --batch 8 --steps 10 --warmup 2 --no-cpu-baseline --no-secondary --profile-events 0 "$@" > $root/gpurun_out/${tag}_c3_prof.json 2> $root/gpurun_out/${tag}_c3_prof.err
csv=$(find /tmp/c3prof -name '*kernel_trace.csv' | head -1)
python3 - "$csv" > $root/gpurun_out/${tag}_c3_trace.txt <<'PY'
import csv, sys, collections
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
agg=collections.defaultdict(list)
for r in rows[len(rows)//3:]:
    agg[r['Kernel_Name'][:70]].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
for k,v in sorted(agg.items(), key=lambda kv:-sum(kv[1])): print("%-72s n=%5d mean %8.2f us total %9.1f" % (k, len(v), sum(v)/len(v), sum(v)))
# a window of consecutive kernels in the middle of a level
mid=len(rows)*2//3
t0=int(rows[mid]['Start_Timestamp'])
for r in rows[mid:mid+12]:
    print("%-60s %9.2f %8.2f" % (r['Kernel_Name'][:60], (int(r['Start_Timestamp'])-t0)/1e3, (int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3))
PY
cat $root/gpurun_out/${tag}_c3_trace.txt
