#!/bin/bash
R=$PWD
python -m pytest tests -m gpu -q 2>&1 | tail -15 > gpurun_out/r02_pytest3.log
NAQS_DEBUG_CLOCKS=1 python tools/clock_probe.py > gpurun_out/r02_clocks.log 2>&1
python tools/train_loop_profile.py > gpurun_out/r02_trainloop.log 2>&1
python tools/step_profile.py >> gpurun_out/r02_trainloop.log 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_train_r02 -o train -- python3 $R/tools/train_loop_profile.py > $R/gpurun_out/prof_train_r02.log 2>&1
B="python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-config4 --pipeline 1"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM --kernel-trace --output-format csv -d $R/gpurun_out/pmc_wait_n2 -o bench -- $B > $R/gpurun_out/pmc5.log 2>&1
rocprofv3 --pmc SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_MFMA --kernel-trace --output-format csv -d $R/gpurun_out/pmc_mix_n2 -o bench -- $B > $R/gpurun_out/pmc6.log 2>&1
cd $R
tail -3 gpurun_out/r02_pytest3.log; cat gpurun_out/r02_trainloop.log | tail -20
