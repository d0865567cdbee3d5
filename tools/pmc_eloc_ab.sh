#!/bin/bash
# instruction / wait counters of the E_loc kernel variants on the headline workload (one batch at a time), per launch
R=$PWD; OUT=$R/gpurun_out/${1:-pmc_eloc_ab}; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for v in 2 3; do
  export NAQS_ELOC_V=$v
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/issue_v$v -o bench -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-config4 --no-train-step --pipeline 1 > $OUT/issue_v$v.log 2>&1
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM --kernel-trace --output-format csv -d $OUT/wait_v$v -o bench -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-config4 --no-train-step --pipeline 1 > $OUT/wait_v$v.log 2>&1
done
cd $R
python3 - <<PY
import csv, glob, collections
for v in (2, 3):
    for kind in ("issue", "wait"):
        acc = collections.defaultdict(list)
        for fn in glob.glob("$OUT/%s_v%d/**/*counter_collection.csv" % (kind, v), recursive=True):
            for r in csv.DictReader(open(fn)):
                if "eloc_kernel" in r["Kernel_Name"]:
                    acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
        print("v%d %s:" % (v, kind), {k: round(sum(x) / len(x)) for k, x in sorted(acc.items())}, "launches", len(next(iter(acc.values()), [])))
PY
