#!/bin/bash
# GPU-box helper: SQ counters of a stand-alone ubench binary (two --pmc passes, kernel trace only)
#   scripts/pmc_ubench.sh <out-dir-name> <binary> [args...]
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
N=$1; shift
D=gpurun_out/r3/pmc_$N
mkdir -p $D
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $D/p1 -- "$@" > $D/p1.log 2>&1 &&
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_VALU SQ_INSTS_MFMA --kernel-trace --output-format csv -d $D/p2 -- "$@" > $D/p2.log 2>&1
python scripts/summarize_pmc.py $D > $D/summary.txt 2>&1
cat $D/summary.txt
# third pass: co-execution of matrix and vector instructions, clock
rocprofv3 --pmc SQ_VALU_MFMA_COEXEC_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_INSTS_LDS --kernel-trace --output-format csv -d $D/p3 -- "$@" > $D/p3.log 2>&1 || tail -5 $D/p3.log
python scripts/summarize_pmc.py $D > $D/summary.txt 2>&1
cat $D/summary.txt
