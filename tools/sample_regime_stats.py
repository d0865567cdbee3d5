"""Per-wave regime histogram of the sampler's binomial draws during training (developer aid; needs a library built with
-DNAQS_SAMPLE_STATS: `bash tools/build_variant.sh build/ab/stats.so -DNAQS_SAMPLE_STATS` and NAQS_HIP_LIB=build/ab/stats.so).
A wave's call of naqs::binomial_group runs the inversion loop when ANY of its lanes is in the small regime (n min(p, q) < 10)
and the BTRS rounds when ANY lane is in the large one — a wave that holds both runs one after the other.
usage: python tools/sample_regime_stats.py [molecule npz] [steps] [warmup]"""
import ctypes, contextlib, io, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "naqs-for-quantum-chemistry_amd")]
import numpy as np, torch
import bench
from naqs_amd import _lib
mol = sys.argv[1] if len(sys.argv) > 1 else "N2"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 300
warm = int(sys.argv[3]) if len(sys.argv) > 3 else 40
lib = _lib.load_library()
raw = ctypes.CDLL(_lib.lib_path())
buf = (ctypes.c_ulonglong * 32)()
res = bench.train_step_probe(torch.device("cuda", 0), mol, steps=steps, warmup=warm)
assert raw.naqs_debug_sample_stats(buf) == 0
a = np.array(list(buf), dtype=np.float64).reshape(2, 4, 4)
print(f"{mol}: mean {res['unique_samples_mean']:.0f} unique samples per step, n_samples = {res['n_samples']:.0e}; counters over warm-up + timed steps "
      f"(the instrumented build is slower: time it with the product build)")
names = ["no draw", "inversion only", "BTRS only", "both (mixed)"]
for g, gname in enumerate(("first split (quad draws)", "second split (pair draws)")):
    tot = a[g, :, 0].sum()
    print(f"  {gname}: {int(tot)} wave-calls")
    for c in range(4):
        n, rounds, exact, isteps = a[g, c]
        if n:
            print(f"    {names[c]:<16} {100 * n / tot:5.1f} % of the wave-calls; per call {rounds / n:4.2f} BTRS rounds, {exact / n:4.2f} exact tests, "
                  f"longest inversion {isteps / n:4.1f} steps")
