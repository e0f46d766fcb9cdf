set -e -o pipefail
root=$PWD; out=$root/gpurun_out; mkdir -p $out
export TMPDIR=/tmp; cd /tmp; rm -rf /tmp/prof_c4
rocprofv3 --kernel-trace --stats -d /tmp/prof_c4/trace -o trace -- python3 $root/bench_c4.py --steps 20 --warmup 3 > $out/c4_under_rocprof.json 2> $out/c4_rocprof.log
db=$(find /tmp/prof_c4/trace -name '*.db' | head -1)
python3 $root/tools/rocpd_summary.py $db $out/c4_kernel_stats.md > /dev/null
rm -rf /tmp/prof_c2
rocprofv3 --kernel-trace --stats -d /tmp/prof_c2/trace -o trace -- python3 $root/bench_c2.py > $out/c2_under_rocprof.json 2>> $out/c4_rocprof.log
db=$(find /tmp/prof_c2/trace -name '*.db' | head -1)
python3 $root/tools/rocpd_summary.py $db $out/c2_kernel_stats.md > /dev/null
