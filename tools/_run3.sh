#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R
for v in old bis2 bis4 old bis2 bis4; do
  if [ $v = new ]; then unset NAQS_HIP_LIB; else export NAQS_HIP_LIB=$R/build/ab/$v.so NAQS_LOADER_LAX=1; fi
  python bench.py --steps 400 --warmup 40 --no-cpu-baseline --no-config4 --no-train-step --pipeline 1 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$v', d['ms_per_step'], d['roofline']['kernel_us'], d['roofline']['other_kernels'][0]['kernel_us'])"
done
