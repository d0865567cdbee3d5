"""Throughput of naqs_logpsi_eloc with D independent batches in flight on D streams (one ham + net handle each)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "naqs-for-quantum-chemistry_amd")]
import torch
sys.argv = ["bench.py"]
import bench
from naqs_amd import hamiltonian, packing
from naqs_amd.hilbert import Encoding, Hilbert
from naqs_amd.wavefunction import NAQSComplex_NADE_orbitals
from naqs_amd.fused import FusedLogPsi
dev = torch.device("cuda", 0)
mol = os.environ.get("MOL", "N2"); M = int(os.environ.get("M", "10000"))
ham_p = packing.load_packed(os.path.join(ROOT, "tests", "golden", f"ham_{mol}.npz"))
keys_np, _, counts_np = bench.make_batch(ham_p, M, 0)
keys = hamiltonian.keys_to_device(keys_np, dev)
hil = Hilbert.get(ham_p.n_qubits, ham_p.n_alpha, ham_p.n_beta, encoding=Encoding.SIGNED)
wf = NAQSComplex_NADE_orbitals(hil, device=dev, qubit_ordering=-1, amp_hidden_size=[64], phase_hidden_size=[512, 512], use_amp_spin_sym=True, use_phase_spin_sym=False, aggregate_phase=False, n_alpha_electrons=ham_p.n_alpha, n_beta_electrons=ham_p.n_beta)
weights = torch.as_tensor(counts_np / counts_np.sum(), dtype=torch.float64, device=dev)
for D in (1, 2, 3):
    hams = [hamiltonian.DevicePauliHamiltonian(ham_p, device=dev) for _ in range(D)]
    nets = [FusedLogPsi(wf) for _ in range(D)]
    streams = [torch.cuda.Stream() for _ in range(D)]
    bufs = [(torch.empty((M, 2), dtype=torch.float32, device=dev), torch.empty((M, 2), dtype=torch.float64, device=dev), torch.zeros(4, dtype=torch.float64, device=dev)) for _ in range(D)]
    for h in hams: h.reserve(M)
    torch.cuda.synchronize()
    def step(k):
        d = k % D
        with torch.cuda.stream(streams[d]):
            nets[d].log_psi_and_local_energy(hams[d], keys, weights=weights, log_psi_out=bufs[d][0], eloc_out=bufs[d][1], sums_out=bufs[d][2])
    for k in range(60): step(k)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    N = 600
    for k in range(N): step(k)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"{mol} M={M} depth {D}: {dt / N * 1e6:.1f} us/step, {M * N / dt / 1e6:.1f} M samples/s, sums={[float(b[2][0]) for b in bufs]}")
