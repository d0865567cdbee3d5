#!/bin/bash
R=$GRAFT_REPO_ROOT; G=$R/gpurun_out/r05a; mkdir -p $G
cd /tmp && export TMPDIR=/tmp
for f in 0 1; do
export NAQS_ELOC_FUSED_SUMS=$f
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $G/prof_serial_$f -o bench -- python3 $R/bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-config4 --no-train-step --pipeline 1 > $G/rocprof_serial_$f.log 2>&1
head -8 $G/prof_serial_$f/bench_kernel_stats.csv | cut -c1-200
done
