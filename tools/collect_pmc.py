#!/usr/bin/env python3
"""Post-process rocprofv3 PMC passes into the counter summaries bench.py reads.

Collect on the GPU box (separate passes, as MI355X_MICROARCH.md prescribes — FETCH_SIZE and WRITE_SIZE do not fit
one pass; never combined with sys/hip traces; the program itself directly after `--`):

    cd /tmp && export TMPDIR=/tmp        # R = the repo
    B="python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-config4 --no-train-step --pipeline 1"
    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_fetch -o bench -- $B
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_write -o bench -- $B
    rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES \
              GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_issue_n2 -o bench -- $B
    (the same issue pass with `--shard rows --molecule Li2O --samples 50000` -> pmc_issue_li2o)

then
    python tools/collect_pmc.py traffic gpurun_out/pmc_fetch gpurun_out/pmc_write profiles/r02_pmc_traffic.json
    python tools/collect_pmc.py issue gpurun_out/pmc_issue_n2 N2_10000 profiles/r02_pmc_issue.json
    python tools/collect_pmc.py issue gpurun_out/pmc_issue_li2o Li2O_50000 profiles/r02_pmc_issue.json

Units/corrections (MI355X_MICROARCH.md, HBM section): FETCH_SIZE/WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports
exactly half of the bytes of a wide (16 B/lane) coalesced stream -> doubled for the streaming kernel (phase_kernel:
16-byte weight loads); other access widths are uncalibrated and reported as counted.  SQ_* instruction counters
are wave-level instruction counts summed over the chip; SQ_BUSY_CYCLES / 32 shader engines = the kernel's duration in
shader cycles (and, over its duration in microseconds, the clock the pass ran at).
"""
import collections
import csv
import glob
import json
import os
import sys

N_SHADER_ENGINES = 32          # MI355X_MICROARCH.md, chip-level parameters
SHORT = ("phase_kernel", "eloc_kernel", "amp_kernel", "prep_kernel", "reduce_kernel")


def short_name(name):
    for s in SHORT:
        if s in name:
            return s
    return None


def rows_of(path):
    files = glob.glob(os.path.join(path, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        raise SystemExit(f"no *counter_collection.csv under {path}")
    for fn in files:
        with open(fn) as f:
            yield from csv.DictReader(f)


def per_kernel(path, counter):
    acc = collections.defaultdict(list)
    for r in rows_of(path):
        if r["Counter_Name"] == counter:
            acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items() if len(v) >= 10}


def source_hash_of(path):
    """naqs_source_hash() of the library the pass ran (the collecting script leaves it in <pass dir>/source_hash.txt):
    bench.py replays a counter file only on the library build it was collected on."""
    fn = os.path.join(path, "source_hash.txt")
    if not os.path.exists(fn):
        raise SystemExit(f"{fn} missing: run  python -c 'from naqs_amd import _lib; print(_lib.load_library()."
                         f"naqs_source_hash().decode())' > {fn}  next to the rocprofv3 pass")
    return open(fn).read().strip()


def full_names(path):
    return sorted({r["Kernel_Name"] for r in rows_of(path) if short_name(r["Kernel_Name"])})


def traffic(fetch_dir, write_dir, out):
    fetch, write = per_kernel(fetch_dir, "FETCH_SIZE"), per_kernel(write_dir, "WRITE_SIZE")
    res = {"_source_hash": source_hash_of(fetch_dir), "_kernels": full_names(fetch_dir)}
    if source_hash_of(write_dir) != res["_source_hash"]:
        raise SystemExit("the FETCH and WRITE passes ran different library builds")
    for name in sorted(set(fetch) | set(write)):
        short = short_name(name)
        if short is None:
            continue
        f_kib, w_kib = fetch.get(name, 0.0), write.get(name, 0.0)
        corr = 2.0 if short == "phase_kernel" else 1.0
        res[short] = {"fetch_kib_counted": f_kib, "write_kib_counted": w_kib, "fetch_correction": corr,
                      "hbm_bytes_per_launch": (f_kib * corr + w_kib) * 1024.0}
    with open(out, "w") as f:
        json.dump(res, f, indent=1, sort_keys=True)
    print(json.dumps(res, indent=1))


def issue(path, workload_key, out):
    """Per launch of each kernel: mean of every counter over the dispatches, the dispatch duration, and the clock."""
    vals = collections.defaultdict(lambda: collections.defaultdict(list))
    dur = collections.defaultdict(dict)
    for r in rows_of(path):
        short = short_name(r["Kernel_Name"])
        if short is None:
            continue
        vals[short][r["Counter_Name"]].append(float(r["Counter_Value"]))
        dur[short][r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3      # us
    res = {}
    for short, counters in vals.items():
        d = sorted(dur[short].values())
        entry = {k: sum(v) / len(v) for k, v in counters.items()}
        entry["launches"] = len(d)
        entry["kernel_us"] = sum(d) / len(d)
        if "SQ_BUSY_CYCLES" in entry and entry["kernel_us"] > 0:
            # SQ_BUSY_CYCLES arrives summed over the chip's 32 shader engines (a one-workgroup kernel, busy on one of
            # them, reports duration x clock once; a full-chip kernel 32 times): per-engine busy cycles = the kernel's
            # duration in shader cycles.  (GRBM_GUI_ACTIVE also covers the dispatch overhead around the kernel's
            # timestamps and is kept only as counted.)
            entry["kernel_cycles"] = entry["SQ_BUSY_CYCLES"] / N_SHADER_ENGINES
            if entry["launches"] and entry.get("SQ_WAVES", 0) >= 256:
                entry["effective_clock_hz"] = entry["kernel_cycles"] / (entry["kernel_us"] * 1e-6)
        res[short] = entry
    try:
        with open(out) as f:
            allres = json.load(f)
    except (OSError, ValueError):
        allres = {}
    h = source_hash_of(path)
    if allres.get("_source_hash", h) != h:
        raise SystemExit(f"{out} holds counters of library build {allres['_source_hash']}, this pass ran {h}: start a new file")
    allres["_source_hash"] = h
    allres["_kernels"] = sorted(set(allres.get("_kernels", [])) | set(full_names(path)))
    allres[workload_key] = res
    with open(out, "w") as f:
        json.dump(allres, f, indent=1, sort_keys=True)
    print(json.dumps({workload_key: res}, indent=1))


if __name__ == "__main__":
    if sys.argv[1] == "traffic":
        traffic(*sys.argv[2:5])
    elif sys.argv[1] == "issue":
        issue(*sys.argv[2:5])
    else:
        raise SystemExit(__doc__)
