#!/bin/bash
# two PMC passes over the 3d BVH render
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; EXTRA="$*"
for pass in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT"; do
  tag=$(echo $pass | cut -c1-12 | tr ' ' '_')
  timeout -s KILL 120 rocprofv3 --pmc $pass --output-format csv -d $R/gpurun_out/pmc3d_$tag -- python3 $R/tools/run3d.py --spp 128 $EXTRA > $R/gpurun_out/pmc3d_$tag.log 2>&1 || exit 1
done
python3 - $R/gpurun_out <<'PY'
import csv, glob, sys, collections
for f in glob.glob(sys.argv[1] + '/pmc3d_*/**/*_counter_collection.csv', recursive=True):
    agg = collections.defaultdict(float); last = None
    rows = [r for r in csv.DictReader(open(f)) if 'render3d' in r['Kernel_Name']]
    last = max(int(r['Dispatch_Id']) for r in rows)
    for r in rows:
        if int(r['Dispatch_Id']) == last: agg[r['Counter_Name']] += float(r['Counter_Value'])
    print(dict(agg))
PY
