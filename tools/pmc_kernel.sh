#!/bin/bash
# usage: tools/pmc_kernel.sh <out-tag> -- <python script and args>     (counter passes only: no trace domains besides the kernel trace)
# Three SQ passes (8 slots each) + one TCC pass; results under gpurun_out/pmc_<tag>_{a,b,c,d}; summarise with tools/pmc_summary.py
tag=$1; shift; shift
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}" && mkdir -p gpurun_out || exit 1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d gpurun_out/pmc_${tag}_a -- python3 "$@" > gpurun_out/pmc_${tag}_a.log 2>&1 || exit 2
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --kernel-trace --output-format csv -d gpurun_out/pmc_${tag}_b -- python3 "$@" > gpurun_out/pmc_${tag}_b.log 2>&1 || exit 3
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVES SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_VMEM SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS --kernel-trace --output-format csv -d gpurun_out/pmc_${tag}_c -- python3 "$@" > gpurun_out/pmc_${tag}_c.log 2>&1 || exit 4
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_${tag}_d -- python3 "$@" > gpurun_out/pmc_${tag}_d.log 2>&1 || exit 5
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_${tag}_e -- python3 "$@" > gpurun_out/pmc_${tag}_e.log 2>&1 || exit 6
echo "pmc passes done: $tag"
