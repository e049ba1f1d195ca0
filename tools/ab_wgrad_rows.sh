#!/bin/bash
for v in ${1:-1 3 1 3}; do
  RUA_TUNE_WGRAD_ROWS=$v timeout -k 10 200 python bench.py --no-cpu-baseline --no-also --steps 60 --warmup 15 2>/dev/null > /tmp/l.json
  python - $v <<'PY'
import json,sys
j=json.loads(open('/tmp/l.json').read().strip().splitlines()[-1])
k=j['roofline']['all_mfma_kernels']
rb={b['block'][:4]:(b['fwd_us'],b['bwd_us']) for b in j['roofline']['resblocks']}
print(sys.argv[1], j['ms_per_step'], {n:k[n]['ms_per_step'] for n in k if 'wgrad' in n}, rb['enc2'], rb['enc3'], rb['dec3'], flush=True)
PY
done
