#!/bin/bash
# A/B of two builds of the library on the box this runs on: tools/ab_lib.sh "<a.so> <b.so> ..." [repeats] [extra bench.py flags]
# ("-" = the in-tree librua_hip.so).  Experiment builds: RUA_BUILD_FLAGS=-D... RUA_BUILD_OUT=<x.so> python -m resunet_a_mltsk_keras_amd.build --force
libs=$1; reps=${2:-2}; shift 2 2>/dev/null
mkdir -p gpurun_out
for r in $(seq $reps); do for l in $libs; do
  if [ "$l" = "-" ]; then unset RUA_LIB_PATH; else export RUA_LIB_PATH=$PWD/$l; fi
  timeout -k 10 150 python bench.py --no-cpu-baseline --no-also --steps 60 --warmup 15 "$@" 2>/dev/null > gpurun_out/ab_line.json || { echo "$l failed"; exit 1; }
  python - "$l" <<'PY' | tee -a gpurun_out/ab.log
import json, sys
j = json.loads(open("gpurun_out/ab_line.json").read().strip().splitlines()[-1])
k = j["roofline"]["all_mfma_kernels"]
print(sys.argv[1], j["ms_per_step"], j["value"], {n: k[n]["ms_per_step"] for n in k if n.startswith(("conv_dmap", "wgrad_dmap"))}, flush=True)
PY
done; done
