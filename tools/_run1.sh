R=$PWD; G=$R/gpurun_out/onecall; mkdir -p $G
timeout 300 python -m pytest tests/test_optimizer_gpu.py -x -q -m gpu --timeout 300 -k "one_call or flat_adam" 2>&1 | tail -2
timeout 200 python tools/sample_clock_probe.py 2>&1 | tail -22
timeout 600 bash tools/train_ab.sh tests/golden/ham_N2.npz NAQS_TRAIN_ONECALL=1 2>&1 | tail -3
