#!/bin/bash
# A/B of library builds on the SAME box for the depth-filter passes (C2, one-GPU C4) and align2D:
#   tools/ab_df.sh build/libA.so android_svo_amd/csrc/libsvo_hip.so ...     (three interleaved rounds)
cd "$(dirname "$0")/.."
for round in 1 2 3; do
  for lib in "$@"; do
    SVO_HIP_LIB="$PWD/$lib" timeout -k 10 300 python tools/ab_df.py 2> gpurun_out/ab_df_tmp.err || { tail -5 gpurun_out/ab_df_tmp.err; exit 1; }
  done
done
