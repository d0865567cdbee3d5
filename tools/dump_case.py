#!/usr/bin/env python3
"""Write a golden E_loc case as a flat binary for tools/abi_smoke.cpp:
   python tools/dump_case.py LiH c1 out.bin"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
mol, tag, out = sys.argv[1:4]
h = np.load(os.path.join(ROOT, "tests", "golden", f"ham_{mol}.npz"))
z = np.load(os.path.join(ROOT, "tests", "golden", f"eloc_{mol}.npz"))
keys, psi, e = z[f"{tag}_keys"], z[f"{tag}_psi_f32"].astype(np.float64), z[f"{tag}_eloc_c128"]
with open(out, "wb") as f:
    np.array([h["n_qubits"], h["n_alpha"], h["n_beta"], len(h["xy"]), len(keys)], np.int64).tofile(f)
    for a in (h["xy"].astype(np.uint64), h["yz"].astype(np.uint64), h["coeff"].astype(np.float64),
              keys.astype(np.uint64), psi, np.stack([e.real, e.imag], -1).astype(np.float64)):
        np.ascontiguousarray(a).tofile(f)
