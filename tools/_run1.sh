timeout 600 python -m pytest tests/test_nade_gpu.py tests/test_optimizer_gpu.py tests/test_abi.py -x -q -m gpu --timeout 300 2>&1 | tail -3
NAQS_TRAIN_MEGA_DEP=0 timeout 300 python -m pytest tests/test_nade_gpu.py -x -q -m gpu --timeout 300 -k "one_launch" 2>&1 | tail -2
timeout 600 bash tools/train_ab.sh tests/golden/ham_N2.npz NAQS_TRAIN_MEGA_DEP=1 NAQS_TRAIN_MEGA_DEP=0 2>&1 | tail -6
timeout 300 bash tools/train_ab.sh tests/golden/ham_H2O.npz NAQS_TRAIN_MEGA_DEP=1 NAQS_TRAIN_MEGA_DEP=0 2>&1 | tail -6
