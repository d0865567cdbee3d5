import os, sys
import numpy as np
ROOT = "/root/repo"
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "naqs-for-quantum-chemistry_amd")]
import torch
from naqs_amd import _lib, hamiltonian as H, packing as P
from oracle import oracle as O
import test_eloc_gpu as T
env = dict(lib=_lib, H=H, P=P, O=O)
N, na, nb = 40, 6, 6
rs = np.random.RandomState(17)
keys = T.random_physical_keys(N, na, nb, 1500, 17)
pairs = rs.randint(0, len(keys), size=(300, 2))
xys = np.unique(np.r_[np.uint64(0), keys[pairs[:, 0]] ^ keys[pairs[:, 1]]])
xy = np.repeat(xys, rs.randint(1, 6, size=len(xys)))
yz = rs.randint(0, 1 << 20, size=len(xy)).astype(np.uint64) | (rs.randint(0, 1 << 20, size=len(xy)).astype(np.uint64) << np.uint64(20))
cf = rs.normal(size=len(xy))
perm = rs.permutation(len(xy)); xy, yz, cf = xy[perm], yz[perm], cf[perm]
lp = T.synth_logpsi(len(keys), 3); psi = np.exp(lp[:, 0] + 1j * lp[:, 1])
want = O.eloc_matrix_free(xy, yz, cf, keys, psi)
for filt in (False, True):
    ham = H.DevicePauliHamiltonian(P.PackedHamiltonian(N, na if filt else -1, nb if filt else -1, xy, yz, cf))
    for stage in ("2", "0"):
        os.environ["NAQS_STAGE"] = stage
        e = T.run_eloc(env, ham, keys, np.stack([psi.real, psi.imag], -1), dtype=torch.float64)
        err = np.abs(e - want) / np.maximum(1, np.abs(want))
        bad = np.flatnonzero(err > 1e-10)
        print(f"filter={filt} stage={stage} Kxy={ham.Kxy} maxerr={err.max():.3e} nbad={len(bad)} first bad rows {bad[:10]}")
        for r in bad[:3]:
            print("   row", r, "got", e[r], "want", want[r])
