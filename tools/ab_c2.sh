#!/bin/bash
# A/B of library builds on the SAME box for config C2 (bench_c2.py): tools/ab_c2.sh build/libA.so build/libB.so ...
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
for round in 1 2 3; do
  for lib in "$@"; do
    SVO_HIP_LIB="$PWD/$lib" timeout -k 10 300 python bench_c2.py --no-cpu-baseline > gpurun_out/ab_c2_tmp.json 2> gpurun_out/ab_c2_tmp.err || { tail -5 gpurun_out/ab_c2_tmp.err; exit 1; }
    python - "$lib" <<'PY'
import json, sys
d = json.loads(open("gpurun_out/ab_c2_tmp.json").read().strip().splitlines()[-1])
print("%-24s align2d %.2f us  depth filter %.2f us" % (sys.argv[1], d["align2d"]["us_per_batch"], d["depth_filter"]["us_per_frame"]))
PY
  done
done
