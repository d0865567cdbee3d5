export NAQS_LOADER_LAX=1
timeout 900 bash tools/train_ab.sh tests/golden/ham_N2.npz - NAQS_HIP_LIB=$PWD/build/ab/libnaqs_head.so - NAQS_HIP_LIB=$PWD/build/ab/libnaqs_head.so 2>&1 | tail -12
timeout 600 bash tools/train_ab.sh tests/golden/ham_H2O.npz - NAQS_HIP_LIB=$PWD/build/ab/libnaqs_head.so 2>&1 | tail -6
