"""bench.py guards: the module imports and its workload helpers are deterministic on the CPU; on the GPU one short run
of the exact driver command line must print ONE parseable JSON line last, with the contract's keys."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REQUIRED = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
            "vs_baseline", "dtype", "data", "config", "roofline"}


def _bench():
    sys.path.insert(0, ROOT)
    import bench
    return bench


def test_bench_imports_and_workload_is_deterministic():
    bench = _bench()
    from naqs_amd import packing
    ham = packing.load_packed(os.path.join(ROOT, "tests", "golden", "ham_N2.npz"))
    k1, lp1, c1 = bench.make_batch(ham, 10000, 0)
    k2, lp2, c2 = bench.make_batch(ham, 10000, 0)
    assert np.array_equal(k1, k2) and np.array_equal(lp1, lp2) and np.array_equal(c1, c2)
    assert len(np.unique(k1)) == 10000 and np.all(np.diff(k1.astype(np.int64)) > 0)
    # B_alg(M) = M (40 + 24 Kxy) + 16 K + 12 Kxy = 91.16 MB for N2 at M = 10 000
    assert bench.algorithmic_bytes(10000, 2239, 378) == 10000 * (40 + 24 * 378) + 16 * 2239 + 12 * 378 == 91160360
    # the log-psi roofline carries both views of the same clock: executed 16-bit flops (`frac`) and the network's own
    # f32 flops (`algorithmic_frac`), both against the dense f16 MFMA peak — the split's 3x separates them
    roof = bench.logpsi_roofline("phase_kernel_ws<3, false, false> [f16x2 + amplitude]", 23.48e-6, 20, 10000, True, 23.48e-6)
    assert roof["algorithmic_flops_per_launch"] == bench.logpsi_flops(20, 10000, True)
    assert roof["algorithmic_frac"] == pytest.approx(roof["algorithmic_flops_per_launch"] / 23.48e-6 / 1e12 / bench.MFMA_BF16_PEAK_TF)
    assert roof["algorithmic_frac"] < roof["frac"] < 3.0 * roof["algorithmic_frac"] * 1.01
    assert roof["isolated"]["algorithmic_frac"] == pytest.approx(roof["algorithmic_frac"])


def test_li2o_batch_and_row_shards():
    """config 4's table: 50 000 distinct physical 30-qubit keys (7 alpha on even bits, 7 beta on odd bits), and the
    padded row shards of the sharded mode tile it."""
    bench = _bench()
    from naqs_amd import packing
    ham = packing.load_packed(os.path.join(ROOT, "tests", "golden", "ham_Li2O.npz"))
    k, lp, c = bench.make_batch(ham, 50000, 0)
    assert len(k) == 50000 and np.all(np.diff(k.astype(np.int64)) > 0) and np.array_equal(k, bench.make_batch(ham, 50000, 0)[0])
    even, odd = sum(1 << q for q in range(0, 30, 2)), sum(1 << q for q in range(1, 30, 2))
    pa = np.array([bin(int(x) & even).count("1") for x in k[:2000]])
    pb = np.array([bin(int(x) & odd).count("1") for x in k[:2000]])
    assert (pa == 7).all() and (pb == 7).all() and int(k.max()) < 2 ** 30
    for M in (1, 7, 10000, 50000, 50001):
        for world in (1, 2, 3, 8):
            sh = [bench.shard_rows(M, r, world) for r in range(world)]
            S = sh[0][0]
            assert all(s_[0] == S for s_ in sh) and S * world >= M
            assert sh[0][1] == 0 and sh[-1][2] == M and all(sh[i][2] == sh[i + 1][1] for i in range(world - 1))


def _run_bench(argv, env_extra, timeout=300):
    env = dict(os.environ, **env_extra)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        if k not in env_extra:
            env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + argv, capture_output=True, text=True,
                       timeout=timeout, env=env)
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    return r, lines


@pytest.mark.parametrize("n,samples", [(2, 10000), (3, 1000)])
def test_gpus_flag_starts_its_own_ranks(n, samples):
    """The driver calls `python bench.py --gpus N`: with no launcher in the environment bench.py must start N ranks
    itself.  Dry run (gloo, no GPU): the N ranks push the sharded step's collectives through the group with the real
    shapes — padded row shards all-gathered into the table, 4 accumulators all-reduced per step — on closed-form
    values that are checked exactly; rank 0's JSON line is the last line of the parent's stdout."""
    r, lines = _run_bench(["--gpus", str(n), "--steps", "4", "--warmup", "1", "--samples", str(samples), "--shard", "rows"],
                          {"NAQS_BENCH_DRY_RUN": "1"})
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads(lines[-1])
    assert d["dry_run"] and d["value"] is None and d["n_gpus"] == d["ranks"] == n and d["backend"] == "gloo"
    assert d["table_ok"] and d["accumulators_ok"] and d["shard"] == "rows"
    assert sum(1 for l in lines if l.lstrip().startswith("{")) == 1        # only rank 0 reports


def test_under_a_launcher_no_second_launch():
    """torchrun convention: WORLD_SIZE/RANK in the environment -> this process IS a rank, nothing is spawned."""
    r, lines = _run_bench(["--gpus", "1", "--steps", "2", "--samples", "64"],
                          {"NAQS_BENCH_DRY_RUN": "1", "WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode == 0, r.stderr[-2000:]
    assert json.loads(lines[-1])["ranks"] == 1


def test_a_failing_rank_fails_the_launch():
    r, lines = _run_bench(["--gpus", "2", "--steps", "2", "--samples", "64", "--molecule", "no-such-molecule"], {}, timeout=600)
    assert r.returncode != 0


@pytest.mark.gpu
def test_row_sharded_table_through_rccl_on_one_gpu():
    """BASELINE config 4 as one rank: Li2O, ONE table of 50 000 keys, the sharded step with its collectives issued
    through an RCCL group of size 1 (all-gather of the log-psi table, all-reduce of the accumulators)."""
    r, lines = _run_bench(["--gpus", "1", "--shard", "rows", "--molecule", "Li2O", "--samples", "50000", "--steps", "20",
                           "--warmup", "3", "--no-cpu-baseline"], {"NAQS_BENCH_FORCE_DIST": "1"}, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads(lines[-1])
    assert REQUIRED <= set(d) and d["scaling"] == "strong" and d["n_gpus"] == 1 and d["value"] > 1e6
    assert "1 RCCL rank(s)" in d["config"]["ranks"] and d["config"]["rows_per_rank"] == 50000
    assert d["config"]["ranks_detail"] == [{"rank": 0, "device": 0, "communicator_world_size": 1}]     # what the communicator saw
    assert np.isfinite(d["config"]["energy"]) and "all-gather" in d["config"]["collectives_per_step"]


@pytest.mark.gpu
def test_two_ranks_on_one_gpu_shard_one_table_to_the_same_energy():
    """The world > 1 code of `--shard rows` on real kernels: two ranks (both on the box's one GPU, collectives over gloo —
    RCCL refuses two ranks per device) evaluate log psi for half of the table each, all-gather it, produce E_loc for
    their row shards and all-reduce the accumulators: the weighted energy of the table must be the single-rank one."""
    argv = ["--shard", "rows", "--molecule", "Li2O", "--samples", "20001", "--steps", "6", "--warmup", "2", "--no-cpu-baseline"]
    one, l1 = _run_bench(["--gpus", "1"] + argv, {}, timeout=900)
    two, l2 = _run_bench(["--gpus", "2"] + argv, {"NAQS_BENCH_ONE_DEVICE": "1", "NAQS_BENCH_BACKEND": "gloo"}, timeout=900)
    assert one.returncode == 0 and two.returncode == 0, one.stderr[-1500:] + two.stderr[-1500:]
    d1, d2 = json.loads(l1[-1]), json.loads(l2[-1])
    assert d2["n_gpus"] == 2 and "2 gloo rank(s)" in d2["config"]["ranks"] and d2["config"]["rows_per_rank"] == 10001
    assert [(r["rank"], r["communicator_world_size"]) for r in d2["config"]["ranks_detail"]] == [(0, 2), (1, 2)]
    e1, e2 = d1["config"]["energy"], d2["config"]["energy"]
    assert np.isfinite(e1) and abs(e1 - e2) < 1e-9 * abs(e1), (e1, e2)


@pytest.mark.gpu
def test_two_ranks_on_one_gpu_weak_mode_runs():
    """The default (independent batches per rank) mode with two ranks: one all-reduce of the accumulators at the end of
    the timed region, then the row-sharded config-4 table across both ranks, one JSON line from rank 0."""
    r, lines = _run_bench(["--gpus", "2", "--steps", "20", "--warmup", "4", "--no-cpu-baseline"],
                          {"NAQS_BENCH_ONE_DEVICE": "1", "NAQS_BENCH_BACKEND": "gloo"}, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads(lines[-1])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["value"] > 1e6 and "cpu_baseline" not in d
    c4 = d["config4_row_sharded"]
    assert "error" not in c4 and c4["rows_per_rank"] == 25000 and "all-gather" in c4["collectives_per_step"]
    assert sum(1 for l in lines if l.lstrip().startswith("{")) == 1


@pytest.mark.gpu
def test_default_line_is_measured_in_this_run():
    """serial figures and the config-4 table are measured by the same process, nothing is replayed from a file except
    the hardware-counter fields, which say so."""
    r, lines = _run_bench(["--gpus", "1", "--steps", "40", "--warmup", "5", "--no-cpu-baseline"], {}, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads(lines[-1])
    assert d["serial"]["measured"].startswith("this run") and "source" not in d["serial"]
    assert d["serial"]["ms_per_step"] > 0 and d["roofline"]["isolated"]["kernel_us"] > 0
    c4 = d["config4_row_sharded"]
    assert "error" not in c4 and c4["scaling"] == "strong" and c4["value"] > 1e6 and "Li2O" in c4["workload"]
    for roof in [d["roofline"]] + d["roofline"]["other_kernels"]:
        if roof.get("traffic") is not None:
            assert roof["traffic_source"]["replayed"] is True
        if "issue" in roof:
            assert roof["issue"]["replayed"] is True and 0 < roof["issue"]["frac"] <= 1.0
    assert "4 distinct key sets" in d["config"]["batches"]


@pytest.mark.gpu
@pytest.mark.parametrize("extra", [["--no-config4"], ["--pipeline", "1", "--no-config4"]])
def test_bench_prints_one_json_line_last(extra):
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "30", "--warmup", "5", "--no-cpu-baseline"] + extra
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    d = json.loads(lines[-1])
    assert REQUIRED <= set(d) and d["n_gpus"] == 1 and d["steps"] == 30 and d["value"] > 1e6
    assert {"bound", "achieved", "peak", "unit", "frac", "traffic"} <= set(d["roofline"])
    assert sum(1 for l in lines if l.lstrip().startswith("{")) == 1
