"""Device-side waits are bounded (csrc/naqs_poll.hpp): a kernel that waits for a word another workgroup publishes gives up
after NAQS_POLL_BUDGET_MS, records the site in the device's error word and leaves without writing results; the host returns
NAQS_ERR_HIP at its next look.  Each case runs in a child process with NAQS_DEBUG_DROP_STORE=<site> (one producer of that site
skips its store) and a 50 ms budget.

The reference has no counterpart: its only failure path in the loop is MaxBatchSizeExceededError
(src/naqs/network/nade.py:39-40, 710-712 -> src/optimizer/energy.py:939-946)."""
import os
import subprocess
import sys
import textwrap

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

PRELUDE = """
import os, sys, ctypes
import numpy as np
import torch
for p in ({root!r}, os.path.join({root!r}, "tests"), os.path.join({root!r}, "naqs-for-quantum-chemistry_amd")):
    sys.path.insert(0, p)
from conftest import GOLDEN, golden
from naqs_amd import _lib, hamiltonian, packing
lib = _lib.load_library()
def expect_timeout(fn, needle):
    try:
        fn()
        torch.cuda.synchronize()
        st = lib.naqs_device_check(0)
        msg = lib.naqs_last_hip_error_string().decode() if st != 0 else ""
    except _lib.NaqsError as e:
        st, msg = -2, str(e)
    assert st == -2, ("no error reported", st)
    assert "timed out" in msg and needle in msg, msg
    assert lib.naqs_device_check(0) == 0          # reported once, then cleared
    print("OK:", msg)
"""


def _run(body, site, budget_ms=50):
    env = dict(os.environ, NAQS_DEBUG_DROP_STORE=str(site), NAQS_POLL_BUDGET_MS=str(budget_ms))
    code = PRELUDE.format(root=ROOT) + NET + textwrap.dedent(body)
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "OK:" in r.stdout
    return r.stdout


NET = """
from test_nade import make_wf
from naqs_amd.fused import FusedLogPsi
z = golden("nade_N2.npz")
hil, wf = make_wf("N2", z, device="cuda")
"""


def test_log_psi_column_split_wait_is_bounded():
    """Small tables take the column-split form of the log-psi kernel (two workgroups per tile; the consumer waits for the
    producer's partial rows)."""
    _run("""
        fused = FusedLogPsi(wf)
        keys = hamiltonian.keys_to_device(z["eval_keys"][:640], wf.device)
        out = torch.full((640, 2), 7.0, dtype=torch.float32, device=wf.device)
        def call():
            fused.log_psi(keys, out=out)
        expect_timeout(call, "column-split")
        assert "SPLIT=1" in fused.last_kernel(), fused.last_kernel()
        assert bool((out[:16] == 7.0).all())                      # the tile whose word was dropped wrote nothing
    """, site=1)


@pytest.mark.parametrize("site,multi,needle", [(2, "1", "look-back"), (3, "4", "several levels per launch")])
def test_sampler_look_back_wait_is_bounded(site, multi, needle):
    """The sampler's workgroups place their children behind those of the preceding workgroups (decoupled look-back); with
    workgroup 0's word missing, the launch gives up, the rest of the call falls through and the call returns an error —
    through the plain call (next entry) and through the polling host of the one-call step alike."""
    _run(f"""
        os.environ["NAQS_SAMPLE_MULTI"] = "{multi}"
        fused = FusedLogPsi(wf)
        def call():
            fused.sample(10 ** 8, seed=5, max_unique=100000)      # (the first draw of a handle runs one level per launch and leaves
            fused.sample(10 ** 8, seed=6, max_unique=100000)      # the level sizes that let the next one fuse levels)
        expect_timeout(call, "{needle}")
    """, site=site)


def test_training_step_waits_are_bounded():
    """vmc_seed_delta_kernel: every workgroup but the first waits for the first one's sums; the one-call step then returns the
    error at its next look at the device (here: the following step)."""
    _run("""
        import tempfile
        from test_optimizer_gpu import make_opt_gpu
        z, hil, wf, opt = make_opt_gpu("N2", tempfile.mkdtemp(), n_samples=4000)       # (tables of <= 4 096 rows: the seed kernel forms the sums)
        assert opt._can_onecall()
        def call():
            opt.run(3, output_freq=10 ** 6)
        expect_timeout(call, "vmc_seed_delta_kernel")
    """, site=4)


def test_repack_scale_chain_wait_is_bounded():
    """naqs_net_set_weights: the f16x2 scales are derived on the device from per-workgroup weight maxima (tagged words)."""
    _run("""
        def call():
            fused = FusedLogPsi(wf)                               # packs the weights
            keys = hamiltonian.keys_to_device(z["eval_keys"][:64], wf.device)
            torch.cuda.synchronize()
            fused.log_psi(keys)                                   # the next entry sees the error word
        expect_timeout(call, "re-pack")
    """, site=5)


def test_a_wait_that_gives_up_fails_its_own_handle_not_its_neighbour():
    """Round 6: the error word is per network handle.  Two handles share the device (the farm's two runs per GPU): the first one's
    column-split kernel gives up (its producer's store is dropped) — the SECOND handle's next calls neither see an error nor lose
    a digit, and the first handle's next call reports its own failure (with one word per device the neighbour's call took the
    error, failed for nothing, and the victim went on with results that had never been written)."""
    _run("""
        fused_a = FusedLogPsi(wf)                                  # created with NAQS_DEBUG_DROP_STORE=1: its producers skip a store
        del os.environ["NAQS_DEBUG_DROP_STORE"]
        hil_b, wf_b = make_wf("N2", z, device="cuda")
        fused_b = FusedLogPsi(wf_b)                                # ... this one's do not
        keys = hamiltonian.keys_to_device(z["eval_keys"][:640], wf.device)
        want = fused_b.log_psi(keys).clone()
        torch.cuda.synchronize()
        out = torch.full((640, 2), 7.0, dtype=torch.float32, device=wf.device)
        fused_a.log_psi(keys, out=out)                             # gives up after the budget, writes nothing for tile 0
        torch.cuda.synchronize()
        assert bool((out[:16] == 7.0).all())
        for _ in range(3):                                         # the neighbour: no error, same numbers
            got = fused_b.log_psi(keys)
            torch.cuda.synchronize()
            assert torch.equal(got, want)
        def call():
            fused_a.log_psi(keys, out=out)                         # the victim: told at its next entry
        expect_timeout(call, "column-split")
    """, site=1)
