#!/usr/bin/env python3
"""Turn the text written by tools/pmc_pass.sh (per-kernel means of rocprofv3 --pmc counters) into JSON.
Usage: tools/pmc_json.py in.txt out.json "<command the passes profiled>" """
import hashlib
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# the sources the profiled kernels are compiled from: bench.py refuses a profile whose hashes differ from the tree's
PROFILED_SOURCES = ("android_svo_amd/csrc/svo_sia.hip", "android_svo_amd/csrc/svo_device_math.h", "android_svo_amd/csrc/Makefile")
# ... and the depth-filter kernels' (kind "df": tools/pmc_c2.sh)
PROFILED_SOURCES_DF = ("android_svo_amd/csrc/svo_depth.hip", "android_svo_amd/csrc/svo_align_device.h",
                       "android_svo_amd/csrc/svo_match_device.h", "android_svo_amd/csrc/svo_device_math.h",
                       "android_svo_amd/csrc/Makefile")


def source_sha256(kind="sia"):
    files = PROFILED_SOURCES_DF if kind == "df" else PROFILED_SOURCES
    return {os.path.basename(f): hashlib.sha256(open(os.path.join(ROOT, f), "rb").read()).hexdigest() for f in files}


def main():
    src, dst, cmd = sys.argv[1], sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else ""
    kind = sys.argv[4] if len(sys.argv) > 4 else "sia"
    kernels = {}
    cur = None
    for line in open(src):
        if line.startswith("#") or not line.strip():
            continue
        m = re.match(r"\s+(\S+)\s+mean/dispatch\s+(\S+)\s+\(n=(\d+)\)", line)
        if m and cur is not None:
            kernels[cur][m.group(1)] = float(m.group(2))
            kernels[cur].setdefault("_dispatches", {})[m.group(1)] = int(m.group(3))
        elif not line.startswith(" "):
            cur = line.strip()
            kernels.setdefault(cur, {})
    # frame pairs per launch of the profiled command: every counter of these kernels is proportional to it (256 -> 1024
    # pairs: x 4.000), so bench.py scales a profile to the launch size it times
    if kind == "df":
        # tools/dfbench_only.py [seeds w h sigma_scale passes]: a stage may be several launches per pass (the alignment stage:
        # three), so a per-pass figure is mean x dispatches / passes (bench_c2.stage_rooflines)
        m = re.search(r"dfbench_only\.py\s+\d+\s+\d+\s+\d+\s+\S+\s+(\d+)", cmd)
        json.dump({"command": cmd, "unit": "counter value per dispatch (mean over the dispatches of the pass)", "passes": int(m.group(1)) if m else 4,
                   "source_sha256": source_sha256("df"), "kernels": kernels}, open(dst, "w"), indent=1)
        print(dst, {k: len(v) - 1 for k, v in kernels.items()})
        return
    m = re.search(r"--batch\s+(\d+)", cmd)
    if m:
        pairs = int(m.group(1))
    else:
        sys.path.insert(0, ROOT)
        import bench
        pairs = bench.DEFAULT_BATCH
    json.dump({"command": cmd, "unit": "counter value per dispatch (mean over the dispatches of the pass)",
               "frame_pairs_per_launch": pairs, "source_sha256": source_sha256(), "kernels": kernels}, open(dst, "w"), indent=1)
    print(dst, {k: len(v) - 1 for k, v in kernels.items()})


if __name__ == "__main__":
    main()
