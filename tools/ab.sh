#!/bin/bash
# A/B of kernel builds on one box: tools/ab.sh build/libA.so build/libB.so ...   (each run: bench.py --steps 10)
for lib in "$@"; do
  for rep in 1 2; do
    SVO_HIP_LIB=$PWD/$lib python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$lib', round(d['value']), round(d['ms_per_step'],3), d['pose_err_vs_cpu_ref']['rot_rad'])" || exit 1
  done
done
