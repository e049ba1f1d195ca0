#!/bin/bash
# Round-5 profile set (run on the GPU box from the repo root): kernel trace + stats of the bench, the PMC counter passes of
# MI355X_MICROARCH.md "rocprofv3 PMC slots" (SQ x3, FETCH_SIZE, WRITE_SIZE: counter passes carry no trace domain besides the
# kernel trace), an unprofiled bench line, an fp32 bench line.  Raw output -> gpurun_out/r05prof; summaries -> profiles/r05/.
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
OUT=gpurun_out/r05prof; mkdir -p $OUT profiles/r05
python3 bench.py > profiles/r05/bench_unprofiled.json 2> $OUT/bench_unprofiled.err || exit 2
echo "unprofiled bench done"
python3 bench.py --dtype f32 --steps 20 --warmup 5 --no-cpu-baseline --no-also > profiles/r05/bench_f32.json 2> $OUT/bench_f32.err || exit 3
echo "fp32 bench done"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --steps 10 --warmup 3 --blocks 1 --no-cpu-baseline --no-also > profiles/r05/bench.json 2> $OUT/bench_trace.err || exit 4
cp $(ls $OUT/trace/*/*_kernel_stats.csv | head -1) profiles/r05/kernel_stats.csv || exit 5
echo "kernel trace done"
tools/pmc_kernel.sh r05 -- bench.py --steps 2 --warmup 1 --blocks 1 --no-cpu-baseline --no-also || exit 6
python3 tools/pmc_summary.py gpurun_out/pmc_r05 profiles/r05/counters_raw.json > $OUT/pmc_summary.txt || exit 7
python3 profiles/summarize_r05.py || exit 8
echo "r05 profile set complete"
