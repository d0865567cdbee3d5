#!/bin/bash
# per-step launch timeline of the one-call training step (rocprofv3 kernel trace + tools/step_timeline.py): N2 and H2O
R=$PWD; G=$R/gpurun_out/timeline; mkdir -p $G
for m in N2 H2O; do python tools/train_loop_profile.py tests/golden/ham_$m.npz 2>&1 | tail -1; done
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $G/n2 -o train -- python3 $R/tools/train_loop_profile.py > $G/n2.log 2>&1
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $G/h2o -o train -- python3 $R/tools/train_loop_profile.py $R/tests/golden/ham_H2O.npz 1000000 300 40 > $G/h2o.log 2>&1
cd $R
for t in n2 h2o; do f=$(find $G/$t -name "train_kernel_trace.csv" | head -1); python3 tools/step_timeline.py $f | tee $G/step_timeline_$t.txt; done
