"""Same network, same seed: does the sampler return the same table every time, under every cut of the tree into launches?
usage: python tools/sampler_repro_probe.py [molecule] [n_samples]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "naqs-for-quantum-chemistry_amd")]
import numpy as np, torch
from naqs_amd import packing
from naqs_amd.hilbert import Encoding, Hilbert
from naqs_amd.wavefunction import NAQSComplex_NADE_orbitals
from naqs_amd.fused import FusedLogPsi
mol = sys.argv[1] if len(sys.argv) > 1 else "Li2O"
n_samples = int(float(sys.argv[2])) if len(sys.argv) > 2 else 10 ** 8
dev = torch.device("cuda", 0)
ham = packing.load_packed(os.path.join(ROOT, "tests", "golden", f"ham_{mol}.npz"))
na, nb = int(ham.n_alpha), int(ham.n_beta)
hil = Hilbert.get(int(ham.n_qubits), na, nb, encoding=Encoding.SIGNED)
torch.manual_seed(3)
wf = NAQSComplex_NADE_orbitals(hil, device=dev, qubit_ordering=-1, amp_hidden_size=[64], phase_hidden_size=[512, 512],
                               use_amp_spin_sym=True, use_phase_spin_sym=False, aggregate_phase=False, n_alpha_electrons=na,
                               n_beta_electrons=nb)
fused = FusedLogPsi(wf)
ref = None
for cfg in (dict(NAQS_SAMPLE_MULTI="1"), dict(NAQS_SAMPLE_MULTI="3"), dict(NAQS_SAMPLE_MULTI="2"), dict(NAQS_SAMPLE_MULTI="1", NAQS_SAMPLE_FUSED="0"),
            dict(NAQS_SAMPLE_MULTI="3", NAQS_SAMPLE_MULTI3_MAX="1000000")):
    for k in ("NAQS_SAMPLE_MULTI", "NAQS_SAMPLE_FUSED", "NAQS_SAMPLE_MULTI3_MAX"):
        os.environ.pop(k, None)
    os.environ.update(cfg)
    bad = 0
    for rep in range(8):
        out = fused.sample(n_samples, seed=77, max_unique=400000)
        torch.cuda.synchronize()
        if ref is None:
            ref = [t.clone() for t in out]
        same = len(out[0]) == len(ref[0]) and all(torch.equal(a, b) for a, b in zip(out, ref))
        bad += 0 if same else 1
        if not same and bad == 1:
            print(f"   {cfg} rep {rep}: M = {len(out[0])} vs {len(ref[0])}")
    print(f"{mol} {cfg}: {8 - bad}/8 draws equal the first table (M = {len(ref[0])})")
