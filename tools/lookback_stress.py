"""Two handles, two threads, two streams of one GPU, sampler calls in a tight loop: does a look-back wait ever expire, with and
without the device's sampler turns (naqs_net_share_device)?   python tools/lookback_stress.py [seconds] [0|1 ...]
NAQS_STRESS_SCALE (default 0.05): factor on the recorded N2 parameters — 0.05 a flat distribution (the widest trees: one level per
launch), 1 the recorded, peaked one (a training step's trees: sample_multi_kernel launches of ~230 one-per-CU workgroups).
Run on the GPU box with a short budget so that an expired wait costs milliseconds, e.g. NAQS_POLL_BUDGET_MS=20."""
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "naqs-for-quantum-chemistry_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("NAQS_POLL_BUDGET_MS", "20")
import torch                                                      # noqa: E402
from conftest import golden                                       # noqa: E402
from test_nade import make_wf                                     # noqa: E402
from naqs_amd._lib import NaqsError                               # noqa: E402
from naqs_amd.fused import FusedLogPsi                            # noqa: E402


def main():
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 10.0
    modes = [int(a) for a in sys.argv[2:]] or [0, 1]
    z = golden("nade_N2.npz")
    torch.manual_seed(7)
    handles = []
    for _ in range(2):
        wf = make_wf("N2", z, device="cuda")[1]
        with torch.no_grad():                                     # a flat distribution: the widest trees the cap allows
            for p in wf.model.parameters():
                p.mul_(float(os.environ.get("NAQS_STRESS_SCALE", "0.05")))
        handles.append(FusedLogPsi(wf))
    for mode in modes:
        for f in handles:
            f.share_device(bool(mode))
        calls, expired, other = [0, 0], [0, 0], []
        stop = time.time() + seconds

        def work(i):
            torch.cuda.set_device(0)
            with torch.cuda.stream(torch.cuda.Stream(device="cuda")):
                k = 0
                while time.time() < stop:
                    k += 1
                    try:
                        handles[i].sample(10 ** 9, seed=1000 * i + k, max_unique=100000)
                        calls[i] += 1
                    except NaqsError as exc:
                        if "wait timed out" in str(exc):
                            expired[i] += 1
                        else:
                            other.append(str(exc))
                            return

        threads = [threading.Thread(target=work, args=(i,)) for i in range(2)]
        t0 = time.time()
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        torch.cuda.synchronize()
        turns = [f.share_device() for f in handles]
        print(f"scale {os.environ.get('NAQS_STRESS_SCALE', '0.05')}, turns {'on ' if mode else 'off'}: {sum(calls)} sampler calls in {time.time() - t0:.1f} s on two threads, "
              f"{sum(expired)} expired waits, waited for a turn {turns}, other errors {other[:2]}", flush=True)


if __name__ == "__main__":
    main()
