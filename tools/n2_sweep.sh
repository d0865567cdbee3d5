#!/bin/bash
# BASELINE config 5: the N2 bond-dissociation sweep with the reference's batch_train_full_mask.sh flags, one geometry
# after the other on ONE GPU, seeds 111 / 222 / 333 (the reference runs five per geometry and reports the best).
# usage (GPU box): bash tools/n2_sweep.sh  ->  gpurun_out/${ROUND:-r03}/n2_sweep.txt
R=$PWD; mkdir -p $R/gpurun_out/${ROUND:-r03}
OUT=$R/gpurun_out/${ROUND:-r03}/n2_sweep.txt
echo "r(A) seed time(s) final_E(Ha) FCI(Ha) error(mHa)" > $OUT
cd naqs-for-quantum-chemistry_amd
for r in 0.75 0.9 1.05 1.2 1.35 1.5 1.65 1.8 1.95 2.1 2.25; do
  for s in 111 222 333; do
    d=/tmp/sweep_${r}_$s
    python -u -m experiments.run -o $d -m ../tests/golden/ham_N2_$r.npz -single_phase -n1 -n_layer 1 -n_hid 64 -n_layer_phase 2 -n_hid_phase 512 -full_mask_psi -s $s -n_train 10000 -output_freq 5000 -save_freq -1 > $d.log 2>&1
    t=$(grep "training time" ${d}_full_mask_psi/summary.txt | awk '{print $NF}')
    e=$(grep "final <E_loc>" ${d}_full_mask_psi/summary.txt | awk '{print $NF}')
    f=$(grep "^FCI" ${d}_full_mask_psi/summary.txt | awk '{print $NF}')
    m=$(grep "^error to FCI" ${d}_full_mask_psi/summary.txt | awk '{print $NF}')
    echo "$r $s $t $e $f $m" >> $OUT
  done
done
cat $OUT
