#!/bin/bash
# Runtime knobs against the tracker chain's per-frame time (GPU box, repo root): kernel arguments in device memory
# (HIP_FORCE_DEV_KERNARG: already the default of this ROCm, =0 costs 20 us per frame), polled completion signals.
set -e
for v in default "HIP_FORCE_DEV_KERNARG=1" "HIP_FORCE_DEV_KERNARG=0" "HSA_ENABLE_INTERRUPT=0" default; do
  if [ "$v" = default ]; then python tools/chain_bench.py --tracker-only > gpurun_out/env_tmp.json 2>/dev/null; else env $v python tools/chain_bench.py --tracker-only > gpurun_out/env_tmp.json 2>/dev/null; fi
  python3 -c "
import json;d=json.load(open('gpurun_out/env_tmp.json'))['hip_tracker'];print('%-28s mean %.4f median %.4f min %.4f  buf %.4f' % ('$v', d['ms_per_frame_total'], d['ms_per_frame_median'], d['ms_per_frame_min'], d['ms_per_frame_image_in_tracker_buffer_median']))"
done
