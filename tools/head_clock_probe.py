"""Where the sampler's head launch (levels 0-3, one workgroup) spends its cycles during training: cycle stamps of thread 0,
averaged over the calls of a training run (developer aid; needs a library built with -DNAQS_HEAD_CLOCKS:
`bash tools/build_variant.sh build/ab/headclk.so -DNAQS_HEAD_CLOCKS` and NAQS_HIP_LIB=build/ab/headclk.so NAQS_LOADER_LAX=1).
usage: python tools/head_clock_probe.py [molecule] [steps] [warmup]"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "naqs-for-quantum-chemistry_amd")]
import numpy as np, torch
import bench
from naqs_amd import _lib
mol = sys.argv[1] if len(sys.argv) > 1 else "N2"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 300
warm = int(sys.argv[3]) if len(sys.argv) > 3 else 40
_lib.load_library()
raw = ctypes.CDLL(_lib.lib_path())
buf = (ctypes.c_longlong * 32)()
res = bench.train_step_probe(torch.device("cuda", 0), mol, steps=steps, warmup=warm)
assert raw.naqs_debug_head_clocks(buf) == 0
a = np.array(list(buf), dtype=np.float64).reshape(8, 4)
calls = a[7, 0]
print(f"{mol}: {int(calls)} head launches; cycles since the kernel's first instruction (100 MHz constant clock x 21 = shader cycles? no: s_memtime counts shader clocks)")
prev = 0.0
for n in range(4):
    t = a[n] / calls
    print(f"  level {n}: begins {t[0]:8.0f} | probabilities +{t[1] - t[0]:7.0f} | two draws +{t[2] - t[1]:7.0f} | compaction +{t[3] - t[2]:7.0f} | level total {t[3] - prev:7.0f}")
    prev = t[3]

mbuf = (ctypes.c_longlong * 96)()
assert raw.naqs_debug_multi_clocks(mbuf) == 0
m = np.array(list(mbuf), dtype=np.float64).reshape(2, 2, 6, 4)
for last in (0, 1):
    for wg in (0, 1):
        calls, nl, nwg = m[last, wg, 5, 0], m[last, wg, 5, 1], m[last, wg, 5, 2]
        if not calls:
            continue
        print(f"  sample_multi_kernel, {'last' if last else 'first'} multi-level launch, {'last' if wg else 'first'} workgroup: {int(calls)} launches, "
              f"{nl / calls:.1f} levels, {nwg / calls:.0f} workgroups")
        prev = 0.0
        for li in range(4):
            t4 = m[last, wg, li] / calls
            if t4[3] == 0:
                continue
            print(f"    level +{li}: begins {t4[0]:8.0f} | probabilities +{t4[1] - t4[0]:7.0f} | two draws +{t4[2] - t4[1]:7.0f} | scan + barrier +{t4[3] - t4[2]:7.0f} | since previous {t4[3] - prev:7.0f}")
            prev = t4[3]
        e = m[last, wg, 4] / calls
        print(f"    look-back done {e[0]:8.0f} (+{e[0] - prev:.0f}) | children written {e[1]:8.0f} (+{e[1] - e[0]:.0f})")
