#!/bin/bash
# A/B of runtime modes of one build on the same box: each argument is an environment assignment ("NAQS_AMP_MODE=1");
# the headline bench runs under each in turn, twice (boxes drift by a few % over a call)
for rep in 1 2; do
for kv in "$@"; do
  echo "== $kv"
  env $kv python bench.py --no-cpu-baseline --no-config4 --no-train-step --steps 400 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['serial']
print(round(d['value']/1e6,1),'M/s', round(d['ms_per_step']*1e3,2),'us/step; serial', round(s['ms_per_step']*1e3,2), 'phase', round(s['logpsi_kernel_us'],2), 'eloc', round(s['eloc_kernel_us'],2))"
done
done
