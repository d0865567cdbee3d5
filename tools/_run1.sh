R=$PWD; G=$R/gpurun_out/onecall; mkdir -p $G
timeout 900 python -m pytest tests/test_nade_gpu.py tests/test_sampler_gpu.py tests/test_optimizer_gpu.py tests/test_variants_gpu.py tests/test_trajectory_gpu.py -x -q -m gpu --timeout 300 2>&1 | tail -8
timeout 600 bash tools/train_ab.sh tests/golden/ham_N2.npz NAQS_TRAIN_MEGA=1 NAQS_TRAIN_MEGA=0 2>&1 | tail -6
timeout 300 bash tools/train_ab.sh tests/golden/ham_H2O.npz NAQS_TRAIN_MEGA=1 NAQS_TRAIN_MEGA=0 2>&1 | tail -6
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $G/prof_train_n2 -o train -- python3 $R/tools/train_loop_profile.py $R/tests/golden/ham_N2.npz 1000000 300 40 > $G/prof_train_n2.log 2>&1
python3 $R/tools/step_timeline.py $G/prof_train_n2/train_kernel_trace.csv
