#!/bin/bash
# Collect SQ / TCC counters for the ALS kernels (separate --pmc passes, as gpurun requires:
# no --pmc together with sys/runtime traces).  Usage on the GPU box:
#   bash profiles/collect_pmc.sh <tag> [bench args]
set -e
TAG=${1:-pmc}; shift || true
ARGS=${@:---steps 2 --warmup 1 --no-cpu-baseline --no-secondary}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
run() {  # name counters...
  local name=$1; shift
  rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d gpurun_out/${TAG}_${name} -- python3 bench.py $ARGS \
     > gpurun_out/${TAG}_${name}.json 2> gpurun_out/${TAG}_${name}.err
}
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVES
run sq2 SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
run sq3 SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM_RD SQ_VALU_MFMA_COEXEC_CYCLES SQ_LDS_UNALIGNED_STALL SQ_INSTS_SMEM SQ_INSTS_VMEM_WR
run tcc1 FETCH_SIZE GRBM_GUI_ACTIVE
run tcc2 WRITE_SIZE TCC_HIT TCC_MISS
