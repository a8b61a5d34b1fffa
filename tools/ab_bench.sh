#!/bin/bash
# A/B timing of two builds of the HIP library on ONE GPU box, alternating, so that box-to-box variance (about +-1 %)
# stays out of the comparison.  Build the candidate in-tree as usual and keep the other build beside it, e.g.
#   cp raytracer_amd/lib/librt_amd.so raytracer_amd/lib/librt_amd_old.so     # before rebuilding
# then, on the box:  bash tools/ab_bench.sh librt_amd_old.so librt_amd.so [bench.py args...]
# Prints the kernel's average launch time (HIP events, rt_stats.kernel_ms) of every run.
set -u
A=$1; B=$2; shift 2
mkdir -p gpurun_out
for i in 1 2 3; do
  for lib in "$A" "$B"; do
    RT_AMD_LIBNAME=$lib timeout -k 10 200 python3 bench.py --configs none --cpu-spp 0 --steps 10 --warmup 2 "$@" > "gpurun_out/ab_${lib}_$i.log" 2>&1 || { echo "run failed: $lib"; tail -n 5 "gpurun_out/ab_${lib}_$i.log"; exit 1; }
    python3 - "$lib" "gpurun_out/ab_${lib}_$i.log" <<'PY'
import json, sys
line = [x for x in open(sys.argv[2]) if x.startswith("{")][-1]
d = json.loads(line)
print(f"{sys.argv[1]:28s} kernel_ms {d['roofline']['kernel_ms']:.3f}  {d['value']:.0f} {d['unit']}")
PY
  done
done
