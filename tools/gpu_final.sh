#!/bin/bash
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r02_smoke.log 2>&1
python bench.py > gpurun_out/r02_bench_final.log 2>&1
python bench.py --shard rows --molecule Li2O --samples 50000 --steps 100 --warmup 10 > gpurun_out/r02_bench_final_li2o.log 2>&1
NAQS_BENCH_FORCE_DIST=1 python bench.py --gpus 1 --steps 100 --warmup 10 --no-cpu-baseline > gpurun_out/r02_bench_final_dist1.log 2>&1
tail -2 gpurun_out/r02_smoke.log; tail -c 300 gpurun_out/r02_bench_final.log; echo; tail -c 300 gpurun_out/r02_bench_final_dist1.log
