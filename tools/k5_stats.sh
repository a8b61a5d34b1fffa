#!/bin/bash
# rocprofv3 kernel stats of the per-pixel kernels behind the trace kernel, at 1080p and 4K (run on the GPU box from the repo root)
export TMPDIR=/tmp
for size in "1920 1080" "3840 2160"; do
  set -- $size
  out="$PWD/gpurun_out/k5_$1"; rm -rf "$out"
  timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$out" -- python3 $PWD/tools/run_k5.py --width $1 --height $2 > "$out.log" 2>&1 || { tail -5 "$out.log"; exit 1; }
  echo "== $1 x $2"
  python3 - "$out" "$1" "$2" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*_kernel_stats.csv", recursive=True)[0]
n = int(sys.argv[2]) * int(sys.argv[3])
for r in csv.DictReader(open(f)):
    name = r["Name"]
    if any(k in name for k in ("resolve2d", "project_xyz", "encode_xyz", "denoise_add", "progressive_merge", "progressive_add")):
        print(f"{name.split('(')[0][:60]:60s} calls {r['Calls']:>3s} avg {float(r['AverageNs'])/1e3:9.1f} us  min {float(r['MinNs'])/1e3:9.1f} us")
PY
done
