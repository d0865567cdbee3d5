#!/bin/bash
R=$PWD
python -m pytest tests -m gpu -q 2>&1 | tail -15 > gpurun_out/r02_pytest4.log
python tools/train_loop_profile.py > gpurun_out/r02_trainloop4.log 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_train_r02b -o train -- python3 $R/tools/train_loop_profile.py > $R/gpurun_out/prof_train_r02b.log 2>&1
cd $R
tail -3 gpurun_out/r02_pytest4.log; tail -2 gpurun_out/r02_trainloop4.log
