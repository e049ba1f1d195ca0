#!/bin/bash
# usage: ab_env.sh "VAR=a VAR=b ..." reps   - alternating bench runs with one environment assignment each
for r in $(seq ${2:-2}); do for kv in $1; do
  env $kv timeout -k 10 200 python bench.py --no-cpu-baseline --no-also --steps 60 --warmup 15 2>/dev/null > /tmp/ab_line.json || { echo "$kv failed"; exit 1; }
  python - "$kv" <<'PY'
import json, sys
j = json.loads(open("/tmp/ab_line.json").read().strip().splitlines()[-1])
e = j["roofline"]["all_entries"]
print(sys.argv[1], j["ms_per_step"], j["value"], {k: e[k]["ms_per_step"] for k in ("rua_wgrad_reduce_batch", "rua_head_fwd_loss_rep", "rua_head_bwd_sums", "rua_bn_bwd", "rua_stem_fwd_stats") if k in e}, flush=True)
PY
done; done
