#!/bin/bash
# the whole r05 profile set in one gpurun call (run from the repo root on the GPU box; gpurun_out/ is what travels back: copy gpurun_out/r05/* into profiles/r05/ and the two json files into profiles/)
bash profiles/collect_r05.sh || exit 1
python3 tools/step_trace.py > profiles/r05/step_trace.txt 2> gpurun_out/step_trace.err || exit 11
python3 profiles/gap_analysis.py $(ls gpurun_out/r05prof/trace/*/*_kernel_trace.csv | head -1) > profiles/r05/launch_gaps.txt 2> gpurun_out/gap.err || exit 12
python3 tools/bench_band128.py > profiles/r05/bench_band128.txt 2> gpurun_out/bb.err || exit 13
BB_LEVEL=4 python3 tools/bench_band128.py > profiles/r05/bench_band128_level4.txt 2> gpurun_out/bb4.err || exit 14
python3 tools/bench_wgrad_rows.py > profiles/r05/bench_wgrad_rows.txt 2> gpurun_out/bwr.err || exit 20
BL_LEVELS=5,6 python3 tools/bench_conv_levels.py > profiles/r05/bench_conv_levels56.txt 2> gpurun_out/bl.err || exit 15
python3 bench.py --force-dp --steps 30 --warmup 10 --no-cpu-baseline --no-also > profiles/r05/bench_force_dp.json 2> gpurun_out/force_dp.err || exit 16
RUA_LIB_PATH=$PWD/scratch/stamps.so python3 tools/band128_phases.py first > profiles/r05/band128_phases.txt 2> gpurun_out/phases.err || exit 18
RUA_LIB_PATH=$PWD/scratch/stamps.so python3 tools/band128_phases.py dgrad >> profiles/r05/band128_phases.txt 2>> gpurun_out/phases.err || exit 19
python3 tools/bf16_gap.py > profiles/r05/bf16_gap.txt 2> gpurun_out/bf16_gap.err || exit 17
mkdir -p gpurun_out/r05 && cp -r profiles/r05/* gpurun_out/r05/ && cp profiles/traffic.json profiles/rocprof_avg.json gpurun_out/r05/
echo "all collected"
