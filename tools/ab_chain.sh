#!/bin/bash
# A/B of library builds on the SAME box for the tracking chain: tools/ab_chain.sh build/libA.so build/libB.so ...
# (each run: tools/chain_bench.py --tracker-only), interleaved three times so that clock drift shows.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
for round in 1 2 3; do
  for lib in "$@"; do
    SVO_HIP_LIB="$PWD/$lib" timeout -k 10 300 python tools/chain_bench.py --tracker-only > gpurun_out/ab_chain_tmp.json 2> gpurun_out/ab_chain_tmp.err || { tail -5 gpurun_out/ab_chain_tmp.err; exit 1; }
    python - "$lib" <<'PY'
import json, sys
d = json.loads(open("gpurun_out/ab_chain_tmp.json").read().strip().splitlines()[-1])["hip_tracker"]
print("%-28s total %.4f  median %.4f  min %.4f ms/frame" % (sys.argv[1], d["ms_per_frame_total"], d["ms_per_frame_median"], d["ms_per_frame_min"]))
PY
  done
done
