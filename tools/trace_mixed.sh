set -e
export TMPDIR=/tmp
root=$PWD
cd /tmp && rm -rf /tmp/mixprof
rocprofv3 --kernel-trace --output-format csv -d /tmp/mixprof -o mix -- python3 $root/tools/mixed_batch_bench.py > $root/gpurun_out/r03b_mixed_prof.json 2> $root/gpurun_out/r03b_mixed_prof.err
csv=$(find /tmp/mixprof -name '*kernel_trace.csv' | head -1)
python3 - "$csv" > $root/gpurun_out/r03b_mixed_trace.txt <<'PY'
import csv, sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows=[r for r in rows if 'sia_fused' in r['Kernel_Name']]
rows.sort(key=lambda r:int(r['Start_Timestamp']))
t0=int(rows[0]['Start_Timestamp'])
for r in rows[-12:]:
    print(r['Kernel_Name'][:60], r.get('Queue_Id'), r.get('Stream_Id'), (int(r['Start_Timestamp'])-t0)/1e3, (int(r['End_Timestamp'])-t0)/1e3, (int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3, r.get('Grid_Size_X') or r.get('Grid_Size'))
PY
cat $root/gpurun_out/r03b_mixed_trace.txt
