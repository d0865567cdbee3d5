#!/usr/bin/env python3
"""Post-process rocprofv3 PMC passes into profiles/<tag>_pmc_traffic.json.

Collect on the GPU box (separate passes, as MI355X_MICROARCH.md prescribes — FETCH_SIZE and WRITE_SIZE do
not fit one pass; never combined with sys/hip traces):

    cd /tmp && export TMPDIR=/tmp
    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_fetch -o bench -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_write -o bench -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline

then `python tools/collect_pmc.py gpurun_out/pmc_fetch gpurun_out/pmc_write profiles/r01_pmc_traffic.json`.

Units/corrections (MI355X_MICROARCH.md, HBM section): the counters are in KiB; on gfx950 FETCH_SIZE reports
exactly half of the bytes of a wide (16 B/lane) coalesced stream -> doubled for the streaming kernel
(phase_kernel: 16-byte weight loads); other access widths are uncalibrated and reported as counted.
"""
import collections
import csv
import json
import sys


def per_kernel(path, counter):
    acc = collections.defaultdict(list)
    with open(path + "/bench_counter_collection.csv") as f:
        for r in csv.DictReader(f):
            if r["Counter_Name"] == counter:
                acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items() if len(v) >= 10}


def main(fetch_dir, write_dir, out):
    fetch, write = per_kernel(fetch_dir, "FETCH_SIZE"), per_kernel(write_dir, "WRITE_SIZE")
    res = {}
    for name in sorted(set(fetch) | set(write)):
        short = "phase_kernel" if "phase_kernel" in name else "eloc_kernel" if "eloc_kernel" in name else \
            "amp_kernel" if "amp_kernel" in name else "prep_kernel" if "prep_kernel" in name else \
            "reduce_kernel" if "reduce_kernel" in name else None
        if short is None:
            continue
        f_kib, w_kib = fetch.get(name, 0.0), write.get(name, 0.0)
        corr = 2.0 if short == "phase_kernel" else 1.0
        res[short] = {"fetch_kib_counted": f_kib, "write_kib_counted": w_kib, "fetch_correction": corr,
                      "hbm_bytes_per_launch": (f_kib * corr + w_kib) * 1024.0}
    with open(out, "w") as f:
        json.dump(res, f, indent=1, sort_keys=True)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main(*sys.argv[1:4])
