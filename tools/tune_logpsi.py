#!/usr/bin/env python3
"""Time the fused log-psi kernels on the GPU box for a few batch sizes / tile heights."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "naqs-for-quantum-chemistry_amd")]
import torch  # noqa: E402
from naqs_amd.fused import FusedLogPsi  # noqa: E402
from naqs_amd.hilbert import Encoding, Hilbert  # noqa: E402
from naqs_amd.wavefunction import NAQSComplex_NADE_orbitals  # noqa: E402

sys.path.insert(0, os.path.join(ROOT, "tools"))
from tune_eloc import random_keys  # noqa: E402

N, na, nb = 20, 7, 7
hil = Hilbert.get(N, na, nb, encoding=Encoding.SIGNED)
wf = NAQSComplex_NADE_orbitals(hil, qubit_ordering=-1, amp_hidden_size=[64], phase_hidden_size=[512, 512],
                               use_amp_spin_sym=True, use_phase_spin_sym=False, aggregate_phase=False,
                               n_alpha_electrons=na, n_beta_electrons=nb, device="cuda")
fused = FusedLogPsi(wf)
for M in [int(a) for a in sys.argv[1:]] or [10000, 12288, 14000]:
    keys = torch.as_tensor(random_keys(N, na, nb, M, 1).view(np.int64), device="cuda")
    out = torch.empty((M, 2), dtype=torch.float32, device="cuda")
    for rb in (1, 2, 3, 4):
        os.environ["NAQS_PHASE_RB"] = str(rb)
        for _ in range(5):
            fused.log_psi(keys, out=out)
        torch.cuda.synchronize()
        fused.prof_enable(50)
        t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
        t0.record()
        for _ in range(50):
            fused.log_psi(keys, out=out)
        t1.record(); torch.cuda.synchronize()
        ms, n = fused.prof_read()
        fused.prof_enable(0)
        flops = 2.0 * M * (18 * 512 + 512 * 512 + 512 * 4)
        print(f"M={M} RB={rb} (BM={16 * rb}, {-(-M // (16 * rb))} WGs): phase {ms / n * 1e3:7.1f} us = {flops / (ms / n * 1e-3) / 1e12:6.1f} TF; "
              f"amp+phase {t0.elapsed_time(t1) / 50 * 1e3:7.1f} us", flush=True)
