"""phase_kernel_wt against phase_kernel_ws and the PyTorch modules at the headline shape (developer aid).
usage: python tools/wt_check.py [M]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "naqs-for-quantum-chemistry_amd"), os.path.join(ROOT, "tests")]
import numpy as np, torch
import bench
from naqs_amd import hamiltonian, packing
from naqs_amd.fused import FusedLogPsi
from naqs_amd.hilbert import Encoding, Hilbert
from naqs_amd.wavefunction import NAQSComplex_NADE_orbitals
M = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
mol = sys.argv[2] if len(sys.argv) > 2 else "N2"
ham_p = packing.load_packed(os.path.join(ROOT, "tests", "golden", f"ham_{mol}.npz"))
keys_np, _, _ = bench.make_batch(ham_p, M, seed=0)
torch.manual_seed(1234)
hil = Hilbert.get(ham_p.n_qubits, ham_p.n_alpha, ham_p.n_beta, encoding=Encoding.SIGNED)
wf = NAQSComplex_NADE_orbitals(hil, device="cuda:0", **bench.published_ansatz(ham_p))
with torch.no_grad():
    for p in wf.model.phase_layers.parameters():
        p.mul_(1.5)
fused = FusedLogPsi(wf)
keys = hamiltonian.keys_to_device(keys_np, "cuda:0")
with torch.no_grad():
    lp_t = wf.log_psi(hil.idx2state(torch.as_tensor(keys_np.astype(np.int64), device="cuda:0")))
res = {}
for mode in ("0", "1"):
    os.environ["NAQS_PHASE_WT"] = mode
    lp = fused.log_psi(keys).clone()
    torch.cuda.synchronize()
    name = fused.last_kernel()
    for _ in range(20):
        fused.log_psi(keys)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(300):
        fused.log_psi(keys)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 300
    res[mode] = lp
    print(f"NAQS_PHASE_WT={mode}: {name}: {dt * 1e6:.2f} us per call; max |log psi - torch| = {float((lp - lp_t).abs().max()):.3e} "
          f"(log|psi| {float((lp[:, 0] - lp_t[:, 0]).abs().max()):.3e}, phase {float((lp[:, 1] - lp_t[:, 1]).abs().max()):.3e})")
print("ws vs wt: log|psi| identical:", bool(torch.equal(res["0"][:, 0], res["1"][:, 0])), " max phase diff", float((res["0"][:, 1] - res["1"][:, 1]).abs().max()))
if os.environ.get("NAQS_DEBUG_CLOCKS") == "1":
    pass
