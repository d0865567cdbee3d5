#!/bin/bash
# GPU run 2: full gpu tests, bench, rocprof stats, PMC passes
R=$PWD
python -m pytest tests -m gpu -q 2>&1 | tail -40 > gpurun_out/r02_pytest2.log
python bench.py > gpurun_out/r02_bench1.log 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -o "SQ_INSTS_[A-Z_0-9]*\|SQ_WAVE_CYCLES\|SQ_BUSY_CYCLES\|SQ_WAVES\b\|GRBM_GUI_ACTIVE\|SQ_ACTIVE_INST_[A-Z_]*\|SQ_WAIT_[A-Z_]*" | sort -u > $R/gpurun_out/r02_counters.txt
B="python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-config4 --pipeline 1"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_fetch -o bench -- $B > $R/gpurun_out/pmc1.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_write -o bench -- $B > $R/gpurun_out/pmc2.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_issue_n2 -o bench -- $B > $R/gpurun_out/pmc3.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_issue_li2o -o bench -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --shard rows --molecule Li2O --samples 50000 > $R/gpurun_out/pmc4.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r02 -o bench -- python3 $R/bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-config4 > $R/gpurun_out/rocprof_r02.log 2>&1
cd $R
tail -3 gpurun_out/r02_pytest2.log
tail -c 600 gpurun_out/r02_bench1.log
