"""Concurrent replicas on one GPU (`experiments.run --farm --per-gpu k`): the reference's protocol is many independent runs
— five seeds per molecule (experiments/bash/naqs/batch_train.sh:11-15), eleven geometries (N2_energy_surface.sh:5-8) — and a
late-training step fills a third of the chip.  The self-launching farm runs the (molecule, seed) jobs through `per_gpu` slots
per device; sharing the GPU must not change a single number of any run."""
import os
import subprocess
import sys

import pytest

from conftest import GOLDEN, PKG

pytestmark = pytest.mark.gpu

FLAGS = ["-single_phase", "-n1", "-n_layer", "1", "-n_hid", "64", "-n_layer_phase", "2", "-n_hid_phase", "512",
         "-n_train", "300", "-output_freq", "1000", "-save_freq", "-1"]


def _energies(run_dir):
    sys.path.insert(0, PKG)
    import pandas as pd
    df = pd.read_pickle(os.path.join(run_dir, "log.pkl"))
    col = [c for c in df.columns if str(c) == "Local energy"][0]
    return df[col].dropna().to_numpy()


def test_replicas_sharing_a_gpu_reproduce_their_solo_runs(tmp_path):
    mols = ",".join(os.path.join(GOLDEN, f"ham_{m}.npz") for m in ("LiH", "H2O"))
    runs = {}
    for k in (1, 2, 4):                 # (k > 2: the farm caps the runtime at two hardware queues, all four jobs start together)
        out = str(tmp_path / f"k{k}")
        subprocess.run([sys.executable, "-m", "experiments.run", "--farm", "--per-gpu", str(k), "--gpus", "1", "--seeds", "111,222",
                        "-m", mols, "-o", out] + FLAGS, cwd=PKG, check=True, stdout=subprocess.DEVNULL, timeout=300)
        names = sorted(os.listdir(out))
        assert names == ["ham_H2O_s111", "ham_H2O_s222", "ham_LiH_s111", "ham_LiH_s222"]
        runs[k] = {n: _energies(os.path.join(out, n)) for n in names}
        assert all(os.path.exists(os.path.join(out, n, "summary.txt")) for n in names)
    for n, e in runs[1].items():
        assert len(e) == 300 and (e == runs[2][n]).all() and (e == runs[4][n]).all(), n      # bit-identical trajectories
    assert not (runs[1]["ham_LiH_s111"] == runs[1]["ham_LiH_s222"]).all()       # and the seeds do differ
