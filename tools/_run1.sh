timeout 300 python -m pytest tests/test_variants_gpu.py -x -q -m gpu --timeout 300 -k "merged or sgd_step" 2>&1 | tail -2
export NAQS_PROFILE_DEFAULT_ANSATZ=1
timeout 900 bash tools/train_ab.sh tests/golden/ham_N2.npz NAQS_AGG_MERGE=0 NAQS_AGG_MERGE=5 NAQS_AGG_MERGE=7 2>&1 | tail -9
