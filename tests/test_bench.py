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


@pytest.mark.gpu
@pytest.mark.parametrize("extra", [[], ["--pipeline", "1"]])
def test_bench_prints_one_json_line_last(extra):
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "30", "--warmup", "5", "--no-cpu-baseline"] + extra
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    d = json.loads(lines[-1])
    assert REQUIRED <= set(d) and d["n_gpus"] == 1 and d["steps"] == 30 and d["value"] > 1e6
    assert {"bound", "achieved", "peak", "unit", "frac", "traffic"} <= set(d["roofline"])
    assert sum(1 for l in lines if l.lstrip().startswith("{")) == 1
