#!/bin/bash
# vector-memory-path counters of the log-psi kernel on the headline workload, one batch at a time (separate passes per block)
R=$PWD; G=$R/gpurun_out/pmc_phase; rm -rf $G; mkdir -p $G
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-config4 --no-train-step --pipeline 1 --no-serial-segment"
pass() { n=$1; shift; timeout 150 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $G/$n -o bench -- $B > $G/$n.log 2>&1 || echo "pass $n failed"; }
pass ta1 TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum GRBM_GUI_ACTIVE
pass ta2 TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum
pass tcp1 TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum
pass tcp2 TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TD_TCP_STALL_CYCLES_sum
pass tcp3 TCP_GATE_EN1_sum TCP_GATE_EN2_sum
pass tcp4 TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum
pass tcp5 TCP_TOTAL_CACHE_ACCESSES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum
pass tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_TAG_STALL_sum
pass td TD_TD_BUSY_sum TD_TC_STALL_sum
pass sq SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM
cd $R
python3 - <<'PY'
import csv, glob, collections, os
G = "gpurun_out/pmc_phase"
for d in sorted(glob.glob(G + "/*/")):
    acc = collections.defaultdict(list)
    for fn in glob.glob(d + "**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(fn)):
            if "phase_kernel" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in sorted(acc.items()):
        print(os.path.basename(d.rstrip("/")), k, "n=%d" % len(v), "avg=%.4g" % (sum(v) / len(v)))
PY
