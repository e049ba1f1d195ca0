#!/bin/bash
# A/B of the data-parallel step on one GPU (one-rank RCCL group): tools/ab_dp.sh "VAR=a VAR=b" [repeats] [extra bench.py flags]
vals=$1; reps=${2:-2}; shift 2 2>/dev/null
for r in $(seq $reps); do for v in $vals; do
  env $v timeout -k 10 200 python bench.py --force-dp --no-cpu-baseline --no-also --steps 60 --warmup 15 "$@" 2>/dev/null > /tmp/dp_line.json || { echo "$v failed"; exit 1; }
  python - "$v" <<'PY'
import json, sys
j = json.loads(open("/tmp/dp_line.json").read().strip().splitlines()[-1])
print(sys.argv[1], j["ms_per_step"], j["value"], j["dp"]["allreduce_exposed_ms"], flush=True)
PY
done; done
