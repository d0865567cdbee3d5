#!/bin/bash
# Round evidence on ONE GPU box: full gpu suite, headline bench (default, driver-like, Li2O row-sharded, RCCL world 1), rocprof
# kernel stats, PMC passes (each stamped with the library's source hash), scaling models, training-step profile, 10 000-step
# training runs with the reference's flags.  usage: ROUND=r06 bash tools/gpu_evidence_pass.sh   -> gpurun_out/$ROUND/
ROUND=${ROUND:-r06}
R=$PWD; G=$R/gpurun_out/$ROUND; mkdir -p $G
HASH=$(python -c "import sys; sys.path.insert(0, 'naqs-for-quantum-chemistry_amd'); from naqs_amd import _lib; print(_lib.load_library().naqs_source_hash().decode())")
echo "library source hash: $HASH" | tee $G/source_hash.txt
timeout 1500 python -m pytest tests -m gpu -q --timeout=300 > $G/pytest.log 2>&1; tail -3 $G/pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $G/smoke.log 2>&1; tail -2 $G/smoke.log
timeout 600 python bench.py > $G/bench.log 2> $G/bench.err
timeout 600 python bench.py --steps 20 --warmup 5 > $G/bench_driver_like.log 2>/dev/null
timeout 600 python bench.py --shard rows --molecule Li2O --samples 50000 --steps 100 --warmup 10 > $G/bench_li2o.log 2>/dev/null
NAQS_BENCH_FORCE_DIST=1 timeout 600 python bench.py --gpus 1 --steps 100 --warmup 10 --no-cpu-baseline > $G/bench_dist1.log 2>/dev/null
timeout 600 python bench.py --emulate-world 1,2,4,8 --molecule Li2O --samples 50000 --steps 100 --warmup 10 > $G/emulate_li2o.log 2>/dev/null
timeout 600 python tools/scaling_model.py tests/golden/ham_N2.npz 300 $G/train_scaling_n2.json > $G/train_scaling_n2.log 2>&1
NAQS_SCALING_PUBLISHED=1 NAQS_SCALING_TRAIN_FIRST=30 timeout 600 python tools/scaling_model.py tests/golden/ham_Li2O.npz 60 $G/train_scaling_li2o_pub.json > $G/train_scaling_li2o_pub.log 2>&1
cd /tmp && export TMPDIR=/tmp
# (--no-serial-segment: the 2000 one-batch-at-a-time steps that precede the timed region would otherwise dominate the averages)
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $G/prof_pipeline2 -o bench -- python3 $R/bench.py --steps 400 --warmup 40 --no-cpu-baseline --no-config4 --no-train-step --no-serial-segment > $G/rocprof_pipeline2.log 2>&1
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $G/prof_serial -o bench -- python3 $R/bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-config4 --no-train-step --pipeline 1 > $G/rocprof_serial.log 2>&1
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $G/prof_li2o -o bench -- python3 $R/bench.py --shard rows --molecule Li2O --samples 50000 --steps 50 --warmup 5 --no-cpu-baseline --no-train-step --pipeline 1 > $G/rocprof_li2o.log 2>&1
B="python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-config4 --no-train-step --pipeline 1"
L="python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-train-step --shard rows --molecule Li2O --samples 50000 --pipeline 1"
ISSUE="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE"
pmc() { d=$G/$1; shift; timeout 300 rocprofv3 --pmc $PMC --kernel-trace --output-format csv -d $d -o bench -- "$@" > $d.log 2>&1; echo $HASH > $d/source_hash.txt; }
PMC="FETCH_SIZE" pmc pmc_fetch $B
PMC="WRITE_SIZE" pmc pmc_write $B
PMC="$ISSUE" pmc pmc_issue_n2 $B
PMC="$ISSUE" pmc pmc_issue_li2o $L
PMC="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" pmc pmc_wait_n2 $B
PMC="SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS" pmc pmc_mfma_n2 $B
# the log-psi kernel's vector-memory path (TA / TCP / TD / TCC, two counters per block and pass)
# (SKIP_PHASE_MEM=1: rounds in which that kernel did not change keep the earlier pass — ten counter passes; 20 s when every pass goes through, 150 s for each one that hangs)
if [ "${SKIP_PHASE_MEM:-0}" != "1" ]; then ( cd $R && bash tools/pmc_phase_mem.sh > $G/pmc_phase_mem.txt 2>&1 ); fi
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $G/prof_train -o train -- python3 $R/tools/train_loop_profile.py > $G/prof_train.log 2>&1
timeout 400 rocprofv3 --kernel-trace --output-format csv -d $G/prof_train_h2o -o train -- python3 $R/tools/train_loop_profile.py $R/tests/golden/ham_H2O.npz 1000000 300 40 > $G/prof_train_h2o.log 2>&1
cd $R
# per-step timelines (launch order, kernel time, idle gap after each kernel) of the one-call training step
for t in train train_h2o; do python3 tools/step_timeline.py $G/prof_$t/train_kernel_trace.csv > $G/step_timeline_$t.txt 2>&1; done
# the one-call step against the call-by-call loop, interleaved on this box
( bash tools/train_ab.sh tests/golden/ham_N2.npz NAQS_TRAIN_ONECALL=1 NAQS_TRAIN_ONECALL=0; bash tools/train_ab.sh tests/golden/ham_H2O.npz NAQS_TRAIN_ONECALL=1 NAQS_TRAIN_ONECALL=0 ) > $G/train_onecall_ab.txt 2>&1
for m in N2 H2O Li2O; do timeout 200 python tools/train_loop_profile.py tests/golden/ham_$m.npz >> $G/train_step_timing.txt 2>&1; done
NAQS_PROFILE_DEFAULT_ANSATZ=1 timeout 200 python tools/train_loop_profile.py >> $G/train_step_timing.txt 2>&1
cd naqs-for-quantum-chemistry_amd
FLAGS="-single_phase -n1 -n_layer 1 -n_hid 64 -n_layer_phase 2 -n_hid_phase 512 -s 111 -n_train 10000 -output_freq 1000 -save_freq -1"
for mol in H2O N2 Li2O; do
  ( time timeout 300 python -u -m experiments.run -o /tmp/train_$mol -m ../tests/golden/ham_$mol.npz $FLAGS ) > $G/train_$mol.log 2>&1
  cp /tmp/train_$mol/summary.txt $G/train_${mol}_summary.txt
done
( time timeout 300 python -u -m experiments.run -o /tmp/train_N2_default -m ../tests/golden/ham_N2.npz -s 111 -n_train 10000 -output_freq 1000 -save_freq -1 ) > $G/train_N2_default.log 2>&1
cp /tmp/train_N2_default/summary.txt $G/train_N2_default_summary.txt
for mol in N2_1.95 N2_2.25; do
  ( time timeout 300 python -u -m experiments.run -o /tmp/train_$mol -m ../tests/golden/ham_$mol.npz $FLAGS -full_mask_psi ) > $G/train_$mol.log 2>&1
  cp /tmp/train_${mol}_full_mask_psi/summary.txt $G/train_${mol}_summary.txt
done
cd $R
# BASELINE config 5 through the farm: 33 runs of 10 000 steps on this one GPU
ROUND=$ROUND bash tools/n2_sweep.sh > /dev/null 2>&1
tail -c 400 $G/bench.log; echo; for m in H2O N2 N2_default Li2O N2_1.95 N2_2.25; do grep "training time\|final <E_loc>\|error to FCI" $G/train_${m}_summary.txt | tr '\n' ' '; echo; done; cat $G/train_step_timing.txt | grep steps; tail -1 $G/n2_sweep.txt
