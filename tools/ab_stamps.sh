#!/bin/bash
# fused-kernel stamps of several -DSVO_STAMPS builds on the same box: tools/ab_stamps.sh build/libA.so build/libB.so
cd "$(dirname "$0")/.."
for lib in "$@"; do
  echo "== $lib"
  SVO_HIP_STAMPS_LIB=$PWD/$lib timeout -k 10 200 python tools/fused_stamps.py 256 2>&1 | python -c "
import sys,re
t=sys.stdin.read()
import numpy as np
nums=[float(x) for x in re.findall(r'-?\d+\.\d*(?:e[+-]?\d+)?|-?\d+', t.split('per wave:')[0].split('exp+mul')[1])]
a=np.array(nums).reshape(-1,5); print(t.split('cycles per evaluation')[0].strip()); print('wave1 mean: eval %.0f  wait %.0f  solve+barrier %.0f  matvec+series %.0f  exp+mul %.0f' % tuple(a.mean(0)))
rest=t.split('per wave:')[1]
print('per wave:'+rest)
"
done
