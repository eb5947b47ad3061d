#!/bin/bash
# GPU-box helper: SQ counters of the conv / attention kernels (separate --pmc passes, kernel-trace only)
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
W=${1:-conv}
mkdir -p gpurun_out/pmc_sq
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d gpurun_out/pmc_sq/p1 -- python scripts/bench_kernels.py $W > gpurun_out/pmc_sq/p1.log 2>&1 &&
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_VALU SQ_INST_CYCLES_SALU --kernel-trace --output-format csv -d gpurun_out/pmc_sq/p2 -- python scripts/bench_kernels.py $W > gpurun_out/pmc_sq/p2.log 2>&1
tail -3 gpurun_out/pmc_sq/p2.log
find gpurun_out/pmc_sq -name "*counter_collection*"
