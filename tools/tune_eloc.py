#!/usr/bin/env python3
"""Sweep the launch shape of the E_loc kernel on the GPU box (kernel time from the HIP-event hook).

    python tools/tune_eloc.py [N2|Li2O ...]
"""
import itertools
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "naqs-for-quantum-chemistry_amd")]
import torch  # noqa: E402
from naqs_amd import hamiltonian, packing  # noqa: E402


def random_keys(N, na, nb, M, seed):
    rs = np.random.RandomState(seed)
    ev, od = np.arange(0, N, 2), np.arange(1, N, 2)
    out = set()
    while len(out) < M:
        a, b = rs.choice(ev, na, replace=False), rs.choice(od, nb, replace=False)
        out.add(int(sum(1 << int(q) for q in a) | sum(1 << int(q) for q in b)))
    return np.sort(np.array(list(out), np.uint64))


def run(mol, M, reps=50):
    hp = packing.load_packed(os.path.join(ROOT, "tests", "golden", f"ham_{mol}.npz"))
    ham = hamiltonian.DevicePauliHamiltonian(hp)
    keys = random_keys(hp.n_qubits, hp.n_alpha, hp.n_beta, M, 1234)
    rs = np.random.RandomState(4321)
    lp = np.stack([rs.normal(-0.5 * np.log(M), 2.0, M), rs.uniform(0, 2 * np.pi, M)], -1).astype(np.float32)
    k = hamiltonian.keys_to_device(keys, ham.device)
    w = torch.as_tensor(lp, device=ham.device)
    out = torch.empty((M, 2), dtype=torch.float64, device=ham.device)
    ham.reserve(M)
    ref = None
    for nt, rpb_mult, stage in itertools.product((256, 512, 1024), (0, 1, 2, 4), (2, 1)):
        os.environ["NAQS_BLOCK"] = str(nt)
        os.environ["NAQS_STAGE"] = str(stage)
        if rpb_mult:
            os.environ["NAQS_ROWS_PER_BLOCK"] = str(rpb_mult * nt // 64)
        else:
            os.environ.pop("NAQS_ROWS_PER_BLOCK", None)
        for _ in range(5):
            ham.local_energy(k, w, kind="log_psi", out=out)
        torch.cuda.synchronize()
        ham.prof_enable(reps)
        for _ in range(reps):
            ham.local_energy(k, w, kind="log_psi", out=out)
        ms, n = ham.prof_read()
        ham.prof_enable(0)
        res = out.cpu().numpy().copy()
        if ref is None:
            ref = res
        ok = np.array_equal(res, ref)
        print(f"{mol} M={M} block={nt} rows/block={'auto' if not rpb_mult else rpb_mult * nt // 64} stage={stage}: "
              f"{ms / n * 1e3:8.2f} us/launch  {M / (ms / n * 1e-3) / 1e6:8.1f} Msamples/s  same={ok}", flush=True)


if __name__ == "__main__":
    mols = sys.argv[1:] or ["N2", "Li2O"]
    for mol in mols:
        run(mol, 10000 if mol != "Li2O" else 50000)
