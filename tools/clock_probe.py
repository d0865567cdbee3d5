"""NAQS_DEBUG_CLOCKS=1 python tools/clock_probe.py: cycle stamps of the phase kernel's stages (workgroup 0)."""
import os, sys
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
ham_p = packing.load_packed(os.path.join(ROOT, "tests", "golden", "ham_N2.npz"))
keys_np, _, _ = bench.make_batch(ham_p, int(os.environ.get("NAQS_PROBE_M", "10000")), 0)
keys = hamiltonian.keys_to_device(keys_np, dev)
hil = Hilbert.get(20, 7, 7, encoding=Encoding.SIGNED)
wf = NAQSComplex_NADE_orbitals(hil, device=dev, qubit_ordering=-1, amp_hidden_size=[64], phase_hidden_size=[512, 512], use_amp_spin_sym=True, use_phase_spin_sym=False, aggregate_phase=False, n_alpha_electrons=7, n_beta_electrons=7)
os.environ.pop("NAQS_DEBUG_CLOCKS", None)
fused = FusedLogPsi(wf)
for _ in range(5): fused.log_psi(keys)
torch.cuda.synchronize()
os.environ["NAQS_DEBUG_CLOCKS"] = "1"
for _ in range(2):
    fused.log_psi(keys); torch.cuda.synchronize(); print("--", file=sys.stderr)
