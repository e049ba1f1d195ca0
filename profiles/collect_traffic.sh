#!/bin/bash
# HBM traffic per kernel launch (MI355X_MICROARCH.md "HBM" + "rocprofv3 PMC slots"): FETCH_SIZE and WRITE_SIZE do not fit
# one pass (3 + 2 of 4 TCC slots), so two counter-only passes over the same short bench run.  No trace domains besides
# the kernel trace.  Output: gpurun_out/pmc_fetch, gpurun_out/pmc_write -> profiles/traffic_summary.py
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
ARGS="--steps 2 --warmup 1 --no-cpu-baseline ${BENCH_ARGS:-}"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_fetch -- python3 bench.py $ARGS > gpurun_out/pmc_fetch.log 2>&1 || exit 2
echo "fetch pass done"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_write -- python3 bench.py $ARGS > gpurun_out/pmc_write.log 2>&1 || exit 3
echo "write pass done"
