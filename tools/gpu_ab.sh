#!/bin/bash
# A/B of several builds of the library on the same box: each argument is a .so (NAQS_HIP_LIB selects it); two rounds,
# because boxes drift by a few % over a call
for rep in 1 2; do
for lib in "$@"; do
  echo "== $lib"
  NAQS_HIP_LIB=$PWD/$lib NAQS_DEBUG_CLOCKS=1 python tools/clock_probe.py 2>&1 | grep "wave 0\|wave 7" | tail -2
  NAQS_HIP_LIB=$PWD/$lib python bench.py --no-cpu-baseline --no-config4 --no-train-step --steps 400 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['serial']
print(round(d['value']/1e6,1),'M/s', round(d['ms_per_step']*1e3,2),'us/step; serial', round(s['ms_per_step']*1e3,2), 'phase', round(s['logpsi_kernel_us'],2), 'eloc', round(s['eloc_kernel_us'],2))"
done
done
