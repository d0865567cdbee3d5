#!/bin/bash
R=$PWD
python -m pytest tests/test_sampler_gpu.py tests/test_variants_gpu.py tests/test_optimizer_gpu.py -m gpu -q -x 2>&1 | tail -8 > gpurun_out/r02_pytest7.log
python tools/train_loop_profile.py > gpurun_out/r02_trainloop7.log 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_train_r02c -o train -- python3 $R/tools/train_loop_profile.py > $R/gpurun_out/prof_train_r02c.log 2>&1
cd $R
tail -3 gpurun_out/r02_pytest7.log; tail -1 gpurun_out/r02_trainloop7.log
python - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/prof_train_r02c/train_kernel_stats.csv')))
steps=220; tot=0; n=0
for r in rows:
    c=int(r['Calls']); t=float(r['TotalDurationNs'])/1e3; tot+=t; n+=c
    if 'sample' in r['Name']: print(r['Name'][:60], round(c/steps,2), round(float(r['AverageNs'])/1e3,1), round(t/steps,1))
print("launches/step", round(n/steps,1), "gpu us/step", round(tot/steps,1))
PY
