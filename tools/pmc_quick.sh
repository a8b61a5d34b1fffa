#!/bin/bash
# One rocprofv3 PMC pass over a short bench run; prints per-dispatch averages for the render kernel.
# Usage: bash tools/pmc_quick.sh <tag> "<counters>" [bench args]
TAG=$1; CTRS=$2; shift 2
OUT="$PWD/gpurun_out/pmc_$TAG"; mkdir -p "$OUT"; export TMPDIR=/tmp
timeout -s KILL 240 rocprofv3 --pmc $CTRS --output-format csv -d "$OUT" -- python3 $PWD/bench.py --steps 2 --warmup 1 --cpu-spp 0 "$@" > "$OUT/log.txt" 2>&1
echo "rc=$?"
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + '/**/*_counter_collection.csv', recursive=True)[0]
agg = collections.defaultdict(float); disp = set()
for r in csv.DictReader(open(f)):
    if 'render' in r['Kernel_Name'] and 'kernel' in r['Kernel_Name']:
        agg[r['Counter_Name']] += float(r['Counter_Value']); disp.add(r['Dispatch_Id'])
print({k: v / max(1, len(disp)) for k, v in agg.items()}, 'dispatches', len(disp))
PY
