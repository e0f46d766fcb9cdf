#!/bin/bash
# A/B of kernel builds on the SAME box: tools/ab_bench.sh build/libA.so build/libB.so ...  (each run: python bench.py --no-secondary)
# prints frames/s and ms per step of every library, interleaved twice so that clock drift shows.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
for round in 1 2; do
  for lib in "$@"; do
    SVO_HIP_LIB="$PWD/$lib" timeout -k 10 300 python bench.py --no-secondary --no-cpu-baseline ${AB_ARGS} > gpurun_out/ab_tmp.json 2> gpurun_out/ab_tmp.err || { tail -5 gpurun_out/ab_tmp.err; exit 1; }
    python - "$lib" <<'PY'
import json, sys
d = json.loads(open("gpurun_out/ab_tmp.json").read().strip().splitlines()[-1])
print("%-32s %10.1f frames/s  %.4f ms/step  pose err %.2e rad %.2e m" % (sys.argv[1], d["value"], d["ms_per_step"],
      d["pose_err_vs_cpu_ref"]["rot_rad"], d["pose_err_vs_cpu_ref"]["trans_m"]))
PY
  done
done
