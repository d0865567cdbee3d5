#!/bin/bash
python -m pytest tests/test_variants_gpu.py tests/test_abi.py -m gpu -q -x 2>&1 | tail -25 > gpurun_out/r02_pytest8.log
tail -25 gpurun_out/r02_pytest8.log
