"""naqs_net_sample (HIP tree sampler) on the GPU.  The reference's sampler draws with numpy's generator on the
host, so parity is statistical: the empirical distribution must be the network's |psi|^2 over the physical
states (chi-square against exact probabilities), with the reference's structural properties (unique, physical,
ascending keys; un-physical draws discarded -> sum(counts) <= n; probs = product of the conditionals)."""
import numpy as np
import pytest
from scipy import stats

from conftest import golden

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def _setup(mol, masking=None):
    from test_nade import make_wf
    from naqs_amd.fused import FusedLogPsi
    from naqs_amd.nade import NadeMasking
    z = golden(f"nade_{mol}.npz")
    hil, wf = make_wf(mol, z, device="cuda", masking=masking or NadeMasking.PARTIAL)
    return hil, wf, FusedLogPsi(wf)


def _exact(hil, fused):
    """probability of every physical state under the network (not normalised over the physical space when the
    last block is un-masked)"""
    keys = torch.as_tensor(np.sort(hil._all_keys()), device="cuda")
    lp = fused.log_psi(keys)
    return keys.cpu().numpy(), np.exp(2.0 * lp[:, 0].double().cpu().numpy())


@pytest.mark.parametrize("mol,n", [("LiH", 2_000_000), ("H2O", 5_000_000), ("N2", 50_000_000)])
def test_distribution_matches_psi_squared(mol, n):
    hil, wf, fused = _setup(mol)
    keys, counts, probs = fused.sample(n, seed=20240607, max_unique=100000)
    k = keys.cpu().numpy()
    c = counts.cpu().numpy()
    assert np.all(np.diff(k) > 0), "unique and ascending (the reference's order for qubit_ordering=-1)"
    assert hil.is_physical(k).all() and (c > 0).all()
    all_keys, p = _exact(hil, fused)
    total = c.sum()
    assert total <= n
    # mass kept = probability of drawing a physical state (un-physical children are dropped, nade.py:695)
    p_phys = p.sum()
    assert abs(total - n * p_phys) < 6 * np.sqrt(n * p_phys * (1 - p_phys)) + 1
    # probs = product of float32 conditionals = exp(2 log|psi|)
    pos = np.searchsorted(all_keys, k)
    assert np.array_equal(all_keys[pos], k)
    assert np.allclose(probs.cpu().numpy(), p[pos], rtol=2e-4, atol=1e-12)
    obs = np.zeros(len(all_keys))
    obs[pos] = c
    expect = p / p_phys * total
    m = expect >= 5
    chi2 = ((obs[m] - expect[m]) ** 2 / expect[m]).sum() + (obs[~m].sum() - expect[~m].sum()) ** 2 / max(expect[~m].sum(), 1e-9)
    assert stats.chi2.sf(chi2, m.sum()) > 1e-4, (chi2, m.sum())


def test_full_masking_keeps_every_draw():
    from naqs_amd.nade import NadeMasking
    hil, wf, fused = _setup("LiH", NadeMasking.FULL)
    keys, counts, probs = fused.sample(10 ** 6, seed=3, max_unique=1000)
    assert counts.sum().item() == 10 ** 6                       # every conditional masked -> nothing to discard
    assert abs(probs.double().sum().item() - 1) < 0.05 or len(keys) < hil.size


def test_deterministic_in_seed_and_matches_torch_sampler_statistically():
    hil, wf, fused = _setup("LiH")
    a = fused.sample(10 ** 6, seed=11, max_unique=1000)
    b = fused.sample(10 ** 6, seed=11, max_unique=1000)
    c = fused.sample(10 ** 6, seed=12, max_unique=1000)
    assert all(torch.equal(x, y) for x, y in zip(a, b))
    assert not torch.equal(a[1], c[1]) if len(a[1]) == len(c[1]) else True
    # the PyTorch sampler of the same network (torch.binomial on device): same distribution
    g = torch.Generator(device="cuda").manual_seed(1)
    states, counts_t, _ = wf.sample(10 ** 6, ret_log_psi=False, generator=g, use_fused=False)
    kt = hil.state2idx(states).squeeze().cpu().numpy().astype(np.int64)
    ka, ca = a[0].cpu().numpy(), a[1].cpu().numpy()
    union = np.union1d(kt, ka)
    oa, ot = np.zeros(len(union)), np.zeros(len(union))
    oa[np.searchsorted(union, ka)] = ca
    ot[np.searchsorted(union, kt)] = counts_t.cpu().numpy()
    m = (oa + ot) >= 10
    # two-sample chi-square
    na, nt = oa.sum(), ot.sum()
    chi2 = ((np.sqrt(nt / na) * oa[m] - np.sqrt(na / nt) * ot[m]) ** 2 / (oa[m] + ot[m])).sum()
    assert stats.chi2.sf(chi2, m.sum() - 1) > 1e-4


def test_overflow_raises_like_the_reference():
    from naqs_amd.nade import MaxBatchSizeExceededError
    hil, wf, fused = _setup("H2O")
    with pytest.raises(MaxBatchSizeExceededError):
        fused.sample(10 ** 7, seed=1, max_unique=50)
    keys, counts, _ = fused.sample(10 ** 7, seed=1, max_unique=441)      # the whole space fits
    assert len(keys) <= 441


def test_large_sample_counts_n2():
    hil, wf, fused = _setup("N2")
    n = 10 ** 12
    keys, counts, probs = fused.sample(n, seed=5, max_unique=100000)
    k = keys.cpu().numpy()
    assert len(k) <= 14400 and np.all(np.diff(k) > 0) and hil.is_physical(k).all()
    total = counts.sum().item()
    assert 0 < total <= n
    # with 1e12 draws the relative frequencies are the probabilities to ~1e-5
    lp = fused.log_psi(keys)
    p = np.exp(2.0 * lp[:, 0].double().cpu().numpy())
    f = counts.double().cpu().numpy() / n
    big = p > 1e-6
    assert np.max(np.abs(f[big] / p[big] - 1)) < 5e-3


def test_thirty_qubit_network_sample_logpsi_and_gradients():
    """Li2O-sized network (15 orbital pairs: the widest amplitude blocks, 28 inputs) with random weights: the sampler's
    structural guarantees, probs == |psi|^2 of the matrix-core log-psi kernel, and the HIP backward against autograd."""
    from naqs_amd.hilbert import Encoding, Hilbert
    from naqs_amd.wavefunction import NAQSComplex_NADE_orbitals
    torch.manual_seed(3)
    hil = Hilbert.get(30, 7, 7, encoding=Encoding.SIGNED)
    wf = NAQSComplex_NADE_orbitals(hil, device="cuda", qubit_ordering=-1, amp_hidden_size=[64], phase_hidden_size=[512, 512],
                                   use_amp_spin_sym=True, use_phase_spin_sym=False, aggregate_phase=False,
                                   n_alpha_electrons=7, n_beta_electrons=7)
    fused = wf.fused()
    keys, counts, probs = fused.sample(60000, seed=9, max_unique=100000)          # (10^7 draws would exceed the cap: near-uniform psi)
    k = keys.cpu().numpy()
    assert len(k) > 1000 and np.all(np.diff(k) > 0) and hil.is_physical(k).all()
    assert 0 < counts.sum().item() <= 60000
    lp = fused.log_psi(keys)
    assert torch.allclose(probs.double(), (2.0 * lp[:, 0].double()).exp(), rtol=3e-4, atol=1e-30)
    # torch modules on the same states
    sub = keys[:: max(1, len(keys) // 2000)].contiguous()
    states = hil.idx2state(sub)
    lp_ref = wf.log_psi(states).reshape(-1, 2)
    assert torch.max(torch.abs(fused.log_psi(sub) - lp_ref.detach())).item() < 1e-4
    g = torch.randn((len(sub), 2), device="cuda", generator=torch.Generator(device="cuda").manual_seed(1)) / len(sub)
    params = list(wf.model.parameters())
    grads_ref = torch.autograd.grad((lp_ref * g).sum(), params, allow_unused=True)
    for p in params:
        p.grad = None
    lp_t, saved = fused.forward_saved(sub)
    fused.backward_saved(saved, g)
    for (name, p), b in zip(wf.model.named_parameters(), grads_ref):
        ref = torch.zeros_like(p) if b is None else b
        scale = float(ref.abs().max()) + 1e-12
        assert float((p.grad - ref).abs().max()) < 2e-4 * scale + 1e-9, (name, float((p.grad - ref).abs().max()), scale)


@pytest.mark.parametrize("mol", ["H2O", "N2"])
def test_launch_fusions_change_nothing(mol, monkeypatch):
    """Every way of cutting the tree into launches draws the same samples: the first levels in one single-workgroup launch
    (sample_head_kernel: five levels / 1024 threads, four / 256, or none), and a level as ONE launch (expand + compaction
    with a look-back scan across workgroups, sample_level_kernel) or as an expand and a scatter launch."""
    hil, wf, fused = _setup(mol)
    from naqs_amd.nade import MaxBatchSizeExceededError
    outs = []
    for head, fuse, multi in (("2", "1", "1"), ("1", "1", "1"), ("0", "1", "1"), ("2", "0", "1"), ("0", "0", "1"), ("1", "1", "3"),
                              ("1", "1", "2"), ("0", "1", "3"), ("2", "1", "2"), ("1", "1", "4"), ("0", "1", "4")):
        monkeypatch.setenv("NAQS_SAMPLE_HEAD", head)
        monkeypatch.setenv("NAQS_SAMPLE_FUSED", fuse)
        monkeypatch.setenv("NAQS_SAMPLE_MULTI", multi)
        # (several levels per launch are chosen from the previous draw's level sizes: the second call is the one that takes them)
        fused.sample(10 ** 8, seed=76, max_unique=100000)
        outs.append(fused.sample(10 ** 8, seed=77, max_unique=100000))
    assert len(outs[0][0]) > 100
    for o in outs[1:]:
        assert all(torch.equal(x, y) for x, y in zip(outs[0], o))
    # a level with more live prefixes than max_unique is an overflow whichever launch it is cut into (under partial masking an
    # inner level may outnumber the final samples, never the reverse): same outcome at caps around the final size
    M = len(outs[0][0])
    for cap in (M - 1, M, M + M // 8, 2 * M):
        res = []
        for multi in ("1", "2", "4"):
            monkeypatch.setenv("NAQS_SAMPLE_HEAD", "1")
            monkeypatch.setenv("NAQS_SAMPLE_FUSED", "1")
            monkeypatch.setenv("NAQS_SAMPLE_MULTI", multi)
            fused.sample(10 ** 8, seed=76, max_unique=100000)
            try:
                res.append(fused.sample(10 ** 8, seed=77, max_unique=cap))
            except MaxBatchSizeExceededError:
                res.append(None)
        assert (res[0] is None) == (res[1] is None) == (res[2] is None), cap
        if res[0] is not None:
            assert all(torch.equal(x, y) for r in res[1:] for x, y in zip(res[0], r))
        if cap == M - 1:
            assert res[0] is None
        if cap == 2 * M and mol == "N2":
            assert res[0] is not None
    for k in ("NAQS_SAMPLE_HEAD", "NAQS_SAMPLE_FUSED", "NAQS_SAMPLE_MULTI"):
        monkeypatch.delenv(k)


def test_sampler_weights_are_counts_over_total():
    hil, wf, fused = _setup("N2")
    keys, counts, probs, weights = fused.sample(10 ** 7, seed=5, max_unique=100000, with_weights=True)
    k2, c2, p2 = fused.sample(10 ** 7, seed=5, max_unique=100000)
    assert torch.equal(keys, k2) and torch.equal(counts, c2) and torch.equal(probs, p2)
    want = counts.double() / counts.sum().double()
    assert weights.dtype == torch.float64 and torch.equal(weights, want) and abs(weights.sum().item() - 1) < 1e-12


# the generator itself on the device: exactly the group draws a tree level runs (binomial_group<4> / <2>, round 5: their own
# log / exp, shared Stirling remainders, one quotient per lane), cases of both regimes interleaved within every wave
GROUP_CASES = [(5, 0.3), (20, 0.4), (100, 0.05), (190, 0.05), (240, 0.05), (1000, 0.013), (1000, 0.5), (50, 0.9), (200, 0.94),
               (5000, 0.37), (10 ** 6, 1e-5), (10 ** 6, 1.2e-5), (10 ** 9, 2e-8), (10 ** 12, 1e-11), (10 ** 12, 0.37), (2 ** 44, 0.5)]


@pytest.mark.parametrize("group", [4, 2])
def test_group_draws_on_the_device_match_the_binomial_pmf(group):
    """chi-square of every case's draws against scipy's exact pmf (bins with >= 5 expected counts; for the two largest n the
    first two moments instead), drawn in ONE launch with the cases interleaved lane group by lane group."""
    import ctypes
    from naqs_amd import _lib
    lib = _lib.load_library()
    C = len(GROUP_CASES)
    per = 300000
    reps = per * C
    n = torch.tensor([c[0] for c in GROUP_CASES], dtype=torch.int64, device="cuda")
    p = torch.tensor([c[1] for c in GROUP_CASES], dtype=torch.float64, device="cuda")
    out = torch.empty(reps, dtype=torch.int64, device="cuda")
    _lib.check(lib.naqs_rng_binomial_device(group, C, n.data_ptr(), p.data_ptr(), ctypes.c_uint64(20260000 + group), reps,
                                            out.data_ptr(), torch.cuda.current_stream().cuda_stream), "naqs_rng_binomial_device")
    torch.cuda.synchronize()
    x_all = out.cpu().numpy().reshape(per, C)
    for ci, (nn, pp) in enumerate(GROUP_CASES):
        x = x_all[:, ci]
        assert x.min() >= 0 and x.max() <= nn, (nn, pp)
        mu, sd = nn * pp, np.sqrt(nn * pp * (1 - pp))
        if nn >= 10 ** 12 and sd > 1e3:
            z = (x.astype(np.float64) - mu) / sd
            assert abs(z.mean()) < 5 / np.sqrt(per) and abs(z.var() - 1) < 5 * np.sqrt(2 / per), (nn, pp, z.mean(), z.var())
            continue
        lo, hi = int(max(0, np.floor(mu - 7 * sd))), int(min(nn, np.ceil(mu + 7 * sd)))
        ks = np.arange(lo, hi + 1)
        expect = stats.binom.pmf(ks, nn, pp) * per
        obs = np.bincount(np.clip(x - lo, 0, hi - lo), minlength=hi - lo + 1)[:hi - lo + 1]
        m = expect >= 5
        chi2 = ((obs[m] - expect[m]) ** 2 / expect[m]).sum()
        assert stats.chi2.sf(chi2, m.sum() - 1) > 1e-5, (group, nn, pp, chi2, m.sum() - 1)
    # a pure function of (seed, i): the same call again gives the same draws
    out2 = torch.empty_like(out)
    _lib.check(lib.naqs_rng_binomial_device(group, C, n.data_ptr(), p.data_ptr(), ctypes.c_uint64(20260000 + group), reps,
                                            out2.data_ptr(), torch.cuda.current_stream().cuda_stream), "naqs_rng_binomial_device")
    assert torch.equal(out, out2)


def test_sampler_calls_of_two_handles_sharing_a_gpu_take_turns_and_draw_the_same():
    """Round 6 (`naqs_net_share_device`; the farm's `--per-gpu 2` sets NAQS_SHARED_GPU=1): two handles driven from two threads on
    two streams of one GPU.  A look-back launch is only certain to end while it is the one such launch in flight, so the handles'
    sampler calls take turns (a host-side turn per device, held until the call's stream has drained).  The draws are those of each
    handle alone, and the calls really were ordered: with both threads sampling in a loop, somebody has had to wait for a turn."""
    import threading
    from test_nade import make_wf
    from naqs_amd.fused import FusedLogPsi
    z = golden("nade_N2.npz")
    handles = [FusedLogPsi(make_wf("N2", z, device="cuda")[1]) for _ in range(2)]
    seeds = [100 + i for i in range(40)]
    alone = [[tuple(t.clone() for t in f.sample(10 ** 8, seed=s, max_unique=100000)) for s in seeds[:4]] for f in handles]
    torch.cuda.synchronize()
    assert [f.share_device() for f in handles] == [0, 0]                      # off by default: nobody waited for anybody
    for f in handles:
        f.share_device(True)
    got, errors = [[], []], []
    start = threading.Barrier(2)

    def work(i):
        try:
            torch.cuda.set_device(0)
            with torch.cuda.stream(torch.cuda.Stream(device="cuda")):
                start.wait(timeout=120)
                for s in seeds:
                    got[i].append(tuple(t.clone() for t in handles[i].sample(10 ** 8, seed=s, max_unique=100000)))
                torch.cuda.current_stream().synchronize()
        except Exception as exc:                                                  # noqa: BLE001 (reported below)
            errors.append(exc)

    threads = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    for i in range(2):
        for a, b in zip(alone[i], got[i][:4]):
            assert all(torch.equal(x, y) for x, y in zip(a, b))
    turns = [f.share_device() for f in handles]
    assert 1 <= sum(turns) <= 2 * len(seeds), turns
    handles[0].share_device(False)                                               # off again: its calls go straight ahead
    before = handles[0].share_device()
    handles[0].sample(10 ** 8, seed=1, max_unique=100000)
    assert handles[0].share_device() == before
