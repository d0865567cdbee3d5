#!/bin/bash
# full gpu suite, headline bench, rocprof stats + PMC passes of the bench, training evidence (10k steps, reference flags)
R=$PWD
python -m pytest tests -m gpu -q 2>&1 | tail -12 > gpurun_out/r02_pytest9.log
python bench.py > gpurun_out/r02_bench9.log 2>&1
python bench.py --shard rows --molecule Li2O --samples 50000 --steps 100 --warmup 10 > gpurun_out/r02_bench9_li2o.log 2>&1
cd /tmp && export TMPDIR=/tmp
# (--no-serial-segment: the 2000 one-batch-at-a-time steps that precede the timed region would otherwise dominate the averages)
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r02 -o bench -- python3 $R/bench.py --steps 400 --warmup 40 --no-cpu-baseline --no-config4 --no-serial-segment > $R/gpurun_out/rocprof_r02.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r02_serial -o bench -- python3 $R/bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-config4 --pipeline 1 > $R/gpurun_out/rocprof_r02s.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r02_li2o -o bench -- python3 $R/bench.py --shard rows --molecule Li2O --samples 50000 --steps 50 --warmup 5 --no-cpu-baseline --pipeline 1 > $R/gpurun_out/rocprof_r02l.log 2>&1
B="python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-config4 --pipeline 1"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_fetch -o bench -- $B > $R/gpurun_out/pmc1.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_write -o bench -- $B > $R/gpurun_out/pmc2.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_issue_n2 -o bench -- $B > $R/gpurun_out/pmc3.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_issue_li2o -o bench -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --shard rows --molecule Li2O --samples 50000 --pipeline 1 > $R/gpurun_out/pmc4.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM --kernel-trace --output-format csv -d $R/gpurun_out/pmc_wait_n2 -o bench -- $B > $R/gpurun_out/pmc5.log 2>&1
rocprofv3 --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS --kernel-trace --output-format csv -d $R/gpurun_out/pmc_mfma_n2 -o bench -- $B > $R/gpurun_out/pmc6.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_train_r02 -o train -- python3 $R/tools/train_loop_profile.py > $R/gpurun_out/prof_train_r02.log 2>&1
cd $R
python tools/train_loop_profile.py > gpurun_out/r02_train_step_n2_timing.txt 2>&1
cd naqs-for-quantum-chemistry_amd
for mol in H2O N2; do
  ( time python -u -m experiments.run -o /tmp/train_$mol -m ../tests/golden/ham_$mol.npz -single_phase -n1 -n_layer 1 -n_hid 64 -n_layer_phase 2 -n_hid_phase 512 -s 111 -n_train 10000 -output_freq 1000 -save_freq -1 ) > ../gpurun_out/r02_train_$mol.log 2>&1
  cp /tmp/train_$mol/summary.txt ../gpurun_out/r02_train_${mol}_summary.txt
done
( time python -u -m experiments.run -o /tmp/train_N2_default -m ../tests/golden/ham_N2.npz -s 111 -n_train 10000 -output_freq 1000 -save_freq -1 ) > ../gpurun_out/r02_train_N2_default.log 2>&1
cp /tmp/train_N2_default/summary.txt ../gpurun_out/r02_train_N2_default_summary.txt
( time python -u -m experiments.run -o /tmp/train_Li2O -m ../tests/golden/ham_Li2O.npz -single_phase -n1 -n_layer 1 -n_hid 64 -n_layer_phase 2 -n_hid_phase 512 -s 111 -n_train 10000 -output_freq 1000 -save_freq -1 ) > ../gpurun_out/r02_train_Li2O.log 2>&1
cp /tmp/train_Li2O/summary.txt ../gpurun_out/r02_train_Li2O_summary.txt
cd $R
tail -3 gpurun_out/r02_pytest9.log; tail -c 400 gpurun_out/r02_bench9.log; for m in H2O N2 N2_default Li2O; do tail -4 gpurun_out/r02_train_$m.log; done
