R=$PWD; G=$R/gpurun_out/onecall; mkdir -p $G
timeout 600 python -m pytest tests/test_sampler_gpu.py -x -q -m gpu --timeout 300 2>&1 | tail -15
timeout 600 python -m pytest tests/test_optimizer_gpu.py tests/test_variants_gpu.py tests/test_trajectory_gpu.py -x -q -m gpu --timeout 300 2>&1 | tail -3
timeout 600 bash tools/train_ab.sh tests/golden/ham_N2.npz NAQS_SAMPLE_MULTI=3 NAQS_SAMPLE_MULTI=2 NAQS_SAMPLE_MULTI=1 2>&1 | tail -9
timeout 300 bash tools/train_ab.sh tests/golden/ham_H2O.npz NAQS_SAMPLE_MULTI=3 NAQS_SAMPLE_MULTI=1 2>&1 | tail -6
