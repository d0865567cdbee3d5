#!/usr/bin/env python3
"""Probe: can the hot path be captured into a hipGraph through torch.cuda.graph?  (run each stage under timeout)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "naqs-for-quantum-chemistry_amd")]
import numpy as np, torch
sys.argv = ["bench.py"]
import bench
from naqs_amd import hamiltonian, packing
from naqs_amd.hilbert import Encoding, Hilbert
from naqs_amd.wavefunction import NAQSComplex_NADE_orbitals
from naqs_amd.fused import FusedLogPsi
which = os.environ.get("PROBE", "all")
dev = torch.device("cuda", 0)
ham_p = packing.load_packed(os.path.join(ROOT, "tests", "golden", "ham_N2.npz"))
ham = hamiltonian.DevicePauliHamiltonian(ham_p, device=dev)
M = 10000
keys_np, lp_np, counts_np = bench.make_batch(ham_p, M, 0)
keys = hamiltonian.keys_to_device(keys_np, dev)
hil = Hilbert.get(20, 7, 7, encoding=Encoding.SIGNED)
wf = NAQSComplex_NADE_orbitals(hil, device=dev, qubit_ordering=-1, amp_hidden_size=[64], phase_hidden_size=[512, 512], use_amp_spin_sym=True, use_phase_spin_sym=False, aggregate_phase=False, n_alpha_electrons=7, n_beta_electrons=7)
fused = FusedLogPsi(wf)
log_psi = torch.empty((M, 2), dtype=torch.float32, device=dev)
weights = torch.as_tensor(counts_np / counts_np.sum(), dtype=torch.float64, device=dev)
eloc = torch.empty((M, 2), dtype=torch.float64, device=dev)
acc = torch.zeros(4, dtype=torch.float64, device=dev)
ham.reserve(M)
def step():
    if which in ("all", "logpsi"):
        fused.log_psi(keys, out=log_psi)
    if which in ("all", "eloc"):
        ham.local_energy(keys, log_psi, kind="log_psi", out=eloc, weights=weights, sums_out=acc)
for _ in range(20): step()
torch.cuda.synchronize()
ref = acc.clone()
print("eager ok", which, flush=True)
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    step()
    torch.cuda.synchronize()
    print("capturing", flush=True)
    with torch.cuda.graph(g, stream=s):
        step()
print("captured", flush=True)
torch.cuda.synchronize()
for _ in range(20): g.replay()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(500): g.replay()
torch.cuda.synchronize()
print(f"graph replay {(time.perf_counter()-t0)/500*1e6:.1f} us/step  same={torch.equal(acc, ref)}", flush=True)
