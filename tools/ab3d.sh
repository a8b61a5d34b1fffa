#!/bin/bash
# A/B timing of two or more builds of the HIP library on the 3d standalone scene (BVH kernel), alternating on one box.
# Usage (on the GPU box): bash tools/ab3d.sh libA.so libB.so [...]   (names under raytracer_amd/lib)
set -u
mkdir -p gpurun_out
for i in 1 2 3; do
  for lib in "$@"; do
    RT_AMD_LIBNAME=$lib timeout -k 10 120 python3 tools/run3d.py --spp 128 > "gpurun_out/ab3d_${lib}_$i.log" 2>&1 || { echo "run failed: $lib"; tail -n 5 "gpurun_out/ab3d_${lib}_$i.log"; exit 1; }
    echo "$lib $(tail -n 1 gpurun_out/ab3d_${lib}_$i.log)"
  done
done
