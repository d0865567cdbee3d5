#!/bin/bash
# BASELINE config 5: the N2 bond-dissociation sweep with the reference's batch_train_full_mask.sh flags on ONE GPU, seeds
# 111 / 222 / 333 (the reference runs five per geometry and reports the best) — through the farm: one process, the 33 runs
# in-process, two at a time (profiles/r04_replicas_per_gpu.txt).
# usage (GPU box): bash tools/n2_sweep.sh  ->  gpurun_out/${ROUND:-r06}/n2_sweep.txt
R=$PWD; mkdir -p $R/gpurun_out/${ROUND:-r06}
OUT=$R/gpurun_out/${ROUND:-r06}/n2_sweep.txt
GEOMS="0.75 0.9 1.05 1.2 1.35 1.5 1.65 1.8 1.95 2.1 2.25"
MOLS=$(for r in $GEOMS; do printf "%s," "$R/tests/golden/ham_N2_$r.npz"; done); MOLS=${MOLS%,}
rm -rf /tmp/sweep
cd naqs-for-quantum-chemistry_amd
t0=$(date +%s%N)
timeout 900 python -u -m experiments.run --farm --per-gpu ${PER_GPU:-2} --gpus 1 --seeds 111,222,333 -m $MOLS -o /tmp/sweep -single_phase -n1 -n_layer 1 -n_hid 64 -n_layer_phase 2 -n_hid_phase 512 -full_mask_psi -n_train 10000 -output_freq 5000 -save_freq -1 > /tmp/sweep.log 2>&1
t1=$(date +%s%N)
cp /tmp/sweep.log $R/gpurun_out/${ROUND:-r06}/n2_sweep_farm.log          # (a run that fails leaves its traceback here and no summary below)
echo "r(A) seed time(s) final_E(Ha) FCI(Ha) error(mHa)" > $OUT
for r in $GEOMS; do
  for s in 111 222 333; do
    f=/tmp/sweep/ham_N2_${r}_s${s}_full_mask_psi/summary.txt
    t=$(grep "training time" $f | awk '{print $NF}')
    e=$(grep "final <E_loc>" $f | awk '{print $NF}')
    c=$(grep "^FCI" $f | awk '{print $NF}')
    m=$(grep "^error to FCI" $f | awk '{print $NF}')
    echo "$r $s $t $e $c $m" >> $OUT
  done
done
echo "33 runs of 10 000 steps in $(( (t1 - t0) / 1000000 )) ms of wall time (one process, --per-gpu ${PER_GPU:-2}; training time per run is measured while two share the GPU)" >> $OUT
grep -c "Traceback" $R/gpurun_out/${ROUND:-r06}/n2_sweep_farm.log | sed 's/^/tracebacks in the farm log: /' >> $OUT
cat $OUT
