#!/bin/bash
# A/B of one tuning key on the box this runs on: tools/ab.sh KEY "v1 v2 ..." [repeats] [extra bench.py flags]   (KEY in either case: _lib.py reads RUA_TUNE_<KEY IN CAPITALS>)
# prints ms_per_step / patches/s of bench.py per value, alternating so that drift shows as spread and not as a difference
key=$1; vals=$2; reps=${3:-2}; shift 3 2>/dev/null
mkdir -p gpurun_out
for r in $(seq $reps); do for v in $vals; do
  env RUA_TUNE_${key^^}=$v timeout -k 10 150 python bench.py --no-cpu-baseline --no-also --steps 60 --warmup 15 "$@" 2>/dev/null > gpurun_out/ab_line.json || { echo "$key=$v failed"; exit 1; }
  python - "$key=$v" <<'PY' | tee -a gpurun_out/ab.log
import json, sys
j = json.loads(open("gpurun_out/ab_line.json").read().strip().splitlines()[-1])
k = j["roofline"]["all_mfma_kernels"]
print(sys.argv[1], j["ms_per_step"], j["value"], {n: k[n]["ms_per_step"] for n in k if n.startswith(("conv_dmap", "wgrad_dmap"))}, flush=True)
PY
done; done
