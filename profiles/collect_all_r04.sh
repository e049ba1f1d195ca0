#!/bin/bash
# the whole r04 profile set in one gpurun call (run from the repo root on the GPU box; gpurun_out/ is what travels back: copy gpurun_out/r04/* into profiles/r04/ and the two json files into profiles/)
bash profiles/collect_r04.sh || exit 1
python3 tools/step_trace.py > profiles/r04/step_trace.txt 2> gpurun_out/step_trace.err || exit 11
python3 profiles/gap_analysis.py $(ls gpurun_out/r04prof/trace/*/*_kernel_trace.csv | head -1) > profiles/r04/launch_gaps.txt 2> gpurun_out/gap.err || exit 12
python3 tools/bench_tail.py > profiles/r04/bench_tail.txt 2> gpurun_out/bench_tail.err || exit 13
python3 bench.py --force-dp --steps 30 --warmup 10 --no-cpu-baseline --no-also > profiles/r04/bench_force_dp.json 2> gpurun_out/force_dp.err || exit 14
python3 tools/bench_wgrad_img.py > profiles/r04/bench_wgrad_img.txt 2> gpurun_out/bwi.err || exit 15
mkdir -p gpurun_out/r04 && cp -r profiles/r04/* gpurun_out/r04/ && cp profiles/traffic.json profiles/rocprof_avg.json gpurun_out/r04/
echo "all collected"
