#!/bin/bash
# A/B of environment settings of ONE build on the same box: each argument is "VAR=value[,VAR2=value2]"; two rounds,
# because boxes drift by a few % over a call.  Example: tools/gpu_ab_env.sh NAQS_PHASE_MODE=1 NAQS_PHASE_MODE=2
for rep in 1 2; do
for setting in "$@"; do
  echo "== $setting"
  (
  IFS=',' read -ra kv <<< "$setting"
  for e in "${kv[@]}"; do export "$e"; done
  NAQS_DEBUG_CLOCKS=1 python tools/clock_probe.py 2>&1 | grep "wave 0\|wave 7" | tail -2
  python bench.py --no-cpu-baseline --no-config4 --no-train-step --steps 400 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['serial']
print(round(d['value']/1e6,1),'M/s', round(d['ms_per_step']*1e3,2),'us/step; serial', round(s['ms_per_step']*1e3,2), 'phase', round(s['logpsi_kernel_us'],2), 'eloc', round(s['eloc_kernel_us'],2))"
  )
done
done
