#!/bin/bash
# rocprofv3 passes for one bench.py workload on the GPU box: kernel trace + stats, then PMC counters in separate runs
# (one block's counters per run, as the hardware collects them).
# Usage (from the repo root on the box): bash tools/profile_r3.sh <tag> [bench args...]
set -u
TAG=$1; shift
OUT="$PWD/gpurun_out/prof_$TAG"
mkdir -p "$OUT"
export TMPDIR=/tmp
BENCH="python3 $PWD/bench.py --steps 5 --warmup 1 --cpu-spp 0 --configs none $*"
run() { # name, rocprof args...
  local name=$1; shift
  echo "== $TAG $name"
  rm -rf "$OUT/$name"   # results of an earlier run under the same tag would be averaged in
  timeout -s KILL 120 rocprofv3 "$@" --output-format csv -d "$OUT/$name" -- $BENCH > "$OUT/$name.log" 2>&1
  local rc=$?
  echo "rc=$rc"; tail -c 300 "$OUT/$name.log"; echo
  return $rc
}
run trace --kernel-trace --stats || exit 1
run pmc_sq1 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE || exit 1
run pmc_sq2 --pmc SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT || exit 1
run pmc_fetch --pmc FETCH_SIZE || exit 1
run pmc_write --pmc WRITE_SIZE || exit 1
# executed floating-point work by type: wave-level instruction counts, the FLOP the hardware books for them (an FMA
# counts 2, per wave instruction), and the share of lanes that were active while vector instructions executed
run pmc_flop1 --pmc SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 || exit 1
run pmc_flop2 --pmc SQ_INSTS_VALU_FLOPS_FP32 SQ_INSTS_VALU_FLOPS_FP64 SQ_INSTS_VALU_FLOPS_FP32_TRANS SQ_INSTS_VALU_FLOPS_FP64_TRANS SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU || exit 1
