"""Concurrent replicas on ONE GPU: the reference's protocol is many independent runs (5 seeds x molecule,
experiments/bash/naqs/batch_train.sh:11-15; 11 geometries, N2_energy_surface.sh:5-8) and a late-training step launches ~80
workgroups on a 256-CU chip — so how many runs per hour does one MI355X deliver with k of them sharing it?

    python tools/replicas_per_gpu.py [out.txt] [n_train] [k,k,...]

Runs the same job list (N2 sweep geometries x seeds, the batch_train_full_mask.sh flags) through
`experiments.run --farm --per-gpu k` (k threads of ONE process per GPU, each with its own HIP stream), times the whole batch (process start-up included: that is what a
user waits for) and checks that every run's trajectory is BIT-IDENTICAL to its k = 1 run (the sampler is deterministic in
the seed and the kernels in their inputs: sharing the GPU must not change a single energy)."""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "naqs-for-quantum-chemistry_amd")
sys.path.insert(0, PKG)                       # (log.pkl's column keys are naqs_amd.optimizer.LogKey members)
out_f = sys.argv[1] if len(sys.argv) > 1 else None
n_train = sys.argv[2] if len(sys.argv) > 2 else "10000"
ks = [int(x) for x in sys.argv[3].split(",")] if len(sys.argv) > 3 else [1, 2]
GEOMS = ["0.75", "1.2", "1.65", "2.25"]
SEEDS = "111,222"
FLAGS = ["-single_phase", "-n1", "-n_layer", "1", "-n_hid", "64", "-n_layer_phase", "2", "-n_hid_phase", "512", "-full_mask_psi",
         "-n_train", n_train, "-output_freq", "5000", "-save_freq", "-1"]
mols = ",".join(os.path.join(ROOT, "tests", "golden", f"ham_N2_{g}.npz") for g in GEOMS)
lines = []


def say(s):
    print(s, flush=True)
    lines.append(s)


def energies(run_dir):
    import pandas as pd
    df = pd.read_pickle(os.path.join(run_dir, "log.pkl"))
    col = [c for c in df.columns if str(c) == "Local energy"][0]
    return df[col].dropna().to_numpy()


def summary_value(run_dir, key):
    for ln in open(os.path.join(run_dir, "summary.txt")):
        if ln.startswith(key):
            return float(ln.split(":")[-1])
    return float("nan")


n_jobs = len(GEOMS) * len(SEEDS.split(","))
ref = {}
if os.environ.get("NAQS_REPLICAS_BASELINE", "1") == "1":
    # the round-3 way (tools/n2_sweep.sh): a fresh process per run, one after the other
    t0 = time.time()
    for g in GEOMS:
        for sd in SEEDS.split(","):
            subprocess.run([sys.executable, "-m", "experiments.run", "-m", os.path.join(ROOT, "tests", "golden", f"ham_N2_{g}.npz"),
                            "-o", f"/tmp/replicas_base/{g}_{sd}", "-s", sd] + FLAGS, cwd=PKG, check=True, stdout=subprocess.DEVNULL, timeout=240)
    wall0 = time.time() - t0
    say(f"baseline, one process per run, one at a time: {wall0:.1f} s  = {n_jobs * 3600 / wall0:.0f} runs/hour")
say(f"{n_jobs} runs (N2 at {', '.join(GEOMS)} A x seeds {SEEDS}; {n_train} steps each, batch_train_full_mask.sh flags) on one GPU")
say("k  wall_s  runs_per_hour  speedup  mean_training_s  identical_to_k1")
base = None
for k in ks:
    out = f"/tmp/replicas_k{k}"
    subprocess.run(["rm", "-rf", out])
    t0 = time.time()
    subprocess.run([sys.executable, "-m", "experiments.run", "--farm", "--per-gpu", str(k), "--gpus", "1", "--seeds", SEEDS,
                    "-m", mols, "-o", out] + FLAGS, cwd=PKG, check=True, stdout=subprocess.DEVNULL, timeout=240)
    wall = time.time() - t0
    same, tt = True, []
    for d in sorted(os.listdir(out)):
        rd = os.path.join(out, d)
        e = energies(rd)
        tt.append(summary_value(rd, "training time"))
        if k == ks[0]:
            ref[d] = e
        else:
            ok = len(e) == len(ref[d]) and bool((e == ref[d]).all())
            if not ok:                                   # where a trajectory leaves its reference run
                n = min(len(e), len(ref[d]))
                first = next((i for i in range(n) if e[i] != ref[d][i]), n)
                say(f"   {d}: {len(e)} vs {len(ref[d])} logged energies, first difference at entry {first}"
                    + (f" ({e[first]!r} vs {ref[d][first]!r})" if first < n else ""))
            same = same and ok
    base = base or wall
    say(f"{k}  {wall:.1f}  {n_jobs * 3600 / wall:.0f}  {base / wall:.2f}  {sum(tt) / len(tt):.2f}  {same if k != ks[0] else '-'}")
if out_f:
    os.makedirs(os.path.dirname(out_f) or ".", exist_ok=True)
    open(out_f, "w").write("\n".join(lines) + "\n")
