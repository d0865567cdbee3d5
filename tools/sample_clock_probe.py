"""NAQS_DEBUG_SAMPLE_CLOCKS=1: cycle stamps of workgroup 0 of every per-level sampler launch (N2 network, trained-like
spread): gap to the previous level's end | U read, weights staged + barrier, expand done, look-back done, children written."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "naqs-for-quantum-chemistry_amd")]
import torch
import bench
from naqs_amd import packing
from naqs_amd.hilbert import Encoding, Hilbert
from naqs_amd.wavefunction import NAQSComplex_NADE_orbitals
from naqs_amd.fused import FusedLogPsi
dev = torch.device("cuda", 0)
ham_p = packing.load_packed(os.path.join(ROOT, "tests", "golden", "ham_N2.npz"))
hil = Hilbert.get(20, 7, 7, encoding=Encoding.SIGNED)
torch.manual_seed(3)
wf = NAQSComplex_NADE_orbitals(hil, device=dev, **bench.published_ansatz(ham_p))
fused = FusedLogPsi(wf)
os.environ.pop("NAQS_DEBUG_SAMPLE_CLOCKS", None)
for i in range(5):
    out = fused.sample(10 ** 6, seed=i, max_unique=100000)
torch.cuda.synchronize()
print("unique", len(out[0]), file=sys.stderr)
os.environ["NAQS_DEBUG_SAMPLE_CLOCKS"] = "1"
for i in range(2):
    fused.sample(10 ** 6, seed=10 + i, max_unique=100000); torch.cuda.synchronize(); print("--", file=sys.stderr)
