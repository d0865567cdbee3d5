"""T(K) = a + b K of the headline step: median wall time of a region of K pipelined steps bracketed by
torch.cuda.synchronize() on both sides, for several K — what a short `--steps` costs beyond the steady-state step."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "naqs-for-quantum-chemistry_amd"))
import torch
import bench
from naqs_amd import hamiltonian, packing
from naqs_amd.hilbert import Encoding, Hilbert
from naqs_amd.wavefunction import NAQSComplex_NADE_orbitals
from naqs_amd.fused import FusedLogPsi

dev = torch.device("cuda", 0)
ham_p = packing.load_packed(os.path.join(ROOT, "tests", "golden", "ham_N2.npz"))
M = 10000
batches = [bench.make_batch(ham_p, M, seed=j) for j in range(4)]
key_sets = [hamiltonian.keys_to_device(b[0], dev) for b in batches]
weight_sets = [torch.as_tensor(b[2] / b[2].sum(), dtype=torch.float64, device=dev) for b in batches]
hil = Hilbert.get(ham_p.n_qubits, ham_p.n_alpha, ham_p.n_beta, encoding=Encoding.SIGNED)
wf = NAQSComplex_NADE_orbitals(hil, device=dev, **bench.published_ansatz(ham_p))
depth = 2
hams = [hamiltonian.DevicePauliHamiltonian(ham_p, device=dev) for _ in range(depth)]
nets = [FusedLogPsi(wf) for _ in range(depth)]
streams = [torch.cuda.Stream(device=dev) for _ in range(depth)]
lps = [torch.empty((M, 2), dtype=torch.float32, device=dev) for _ in range(depth)]
els = [torch.empty((M, 2), dtype=torch.float64, device=dev) for _ in range(depth)]
acc = torch.zeros((4096, 4), dtype=torch.float64, device=dev)
for h in hams: h.reserve(M)
n = [0]
def step():
    i = n[0]; d = i % depth
    with torch.cuda.stream(streams[d]):
        nets[d].log_psi_and_local_energy(hams[d], key_sets[i % 4], weights=weight_sets[i % 4], log_psi_out=lps[d], eloc_out=els[d], sums_out=acc[i % 4096])
    n[0] += 1
if os.environ.get("NAQS_PROBE_LEAN") == "1":         # the step through the C ABI with prepared arguments (bench.py's way)
    import ctypes
    from naqs_amd import _lib as L
    call = L.load_library().naqs_logpsi_eloc
    vp = ctypes.c_void_p
    prep = [[(nets[d]._h, hams[d]._h, M, vp(key_sets[k].data_ptr()), vp(weight_sets[k].data_ptr()), vp(lps[d].data_ptr()), vp(els[d].data_ptr()),
              vp(streams[d].cuda_stream)) for k in range(4)] for d in range(depth)]
    acc0 = acc.data_ptr()
    def step():
        i = n[0]; d = i % depth
        a = prep[d][i % 4]
        assert call(a[0], a[1], a[2], a[3], a[4], a[5], a[6], vp(acc0 + 32 * (i % 4096)), a[7]) == 0
        n[0] += 1
for _ in range(300): step()
torch.cuda.synchronize()
for K in (0, 1, 2, 4, 8, 20, 40, 100):
    ts = []
    for rep in range(30):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(K): step()
        t_enq = time.perf_counter() - t0
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0, t_enq))
    a = np.median([t[0] for t in ts]) * 1e6; e = np.median([t[1] for t in ts]) * 1e6
    first = ts[0][0] * 1e6
    print(f"K={K:4d}: region {a:8.1f} us  ({a / max(K, 1):6.2f} us/step; first repetition {first / max(K, 1):6.2f})   host enqueue {e:8.1f} us ({e / max(K, 1):5.1f} per step)")
