#!/bin/bash
# Kernel times of every bench.py configuration for two builds of the library, alternated on one box:
#   bash tools/ab_configs.sh libA.so libB.so      (names under raytracer_amd/lib)
set -u
mkdir -p gpurun_out
for i in 1 2; do
  for lib in "$@"; do
    RT_AMD_LIBNAME=$lib timeout -k 10 300 python3 bench.py --cpu-spp 0 --steps 5 --warmup 1 > "gpurun_out/abc_${lib}_$i.json" 2> "gpurun_out/abc_${lib}_$i.err" || { echo "run failed: $lib"; tail -n 5 "gpurun_out/abc_${lib}_$i.err"; exit 1; }
    python3 - "$lib" "gpurun_out/abc_${lib}_$i.json" <<'PY'
import json, sys
d = json.loads([x for x in open(sys.argv[2]) if x.startswith("{")][-1])
ms = [d["roofline"]["kernel_ms"]] + [c["roofline"]["kernel_ms"] for c in d.get("configs", [])]
print(f"{sys.argv[1]:24s} " + "  ".join(f"{m:7.2f}" for m in ms))
PY
  done
done
