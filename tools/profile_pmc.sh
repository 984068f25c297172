#!/bin/bash
# Collects hardware counters for the walk kernel: one rocprofv3 --pmc run per counter group
# (gfx950 limits: 8 SQ, 4 TCC per pass; FETCH_SIZE takes 3 TCC slots, WRITE_SIZE 2).
# Usage (on the GPU box, from the repo root):  bash tools/profile_pmc.sh <tag> [bench args...]
set -u
TAG=${1:-pmc}; shift || true
ARGS=${@:---photons 4000000 --steps 1 --warmup 0 --no-cpu}
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
pass() {
  name=$1; shift
  timeout -s KILL 180 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/$name" -- python3 bench.py $ARGS > "$OUT/$name.log" 2>&1 || { echo "pass $name failed"; tail -5 "$OUT/$name.log"; return 1; }
  echo "pass $name ok"
}
pass sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY &&
pass sq2 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_SALU SQ_INSTS_VMEM_RD &&
pass sq3 SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_ACTIVE_INST_ANY &&
pass sq4 SQ_INSTS_VALU_CVT SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_INSTS_LDS_ATOMIC SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_CYCLES SQ_INSTS_FLAT &&
pass tcc1 FETCH_SIZE TCC_HIT_sum &&
pass tcc2 WRITE_SIZE TCC_MISS_sum TCC_EA0_ATOMIC_sum &&
pass grbm GRBM_GUI_ACTIVE GRBM_COUNT
