import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "naqs-for-quantum-chemistry_amd")]
import numpy as np, torch
from collections import Counter
keys = [torch.sort(torch.randperm(14400, device="cuda")[:2000])[0].to(torch.int64) for _ in range(256)]
torch.cuda.synchronize()
for rep in range(3):
    t0 = time.perf_counter(); cat = torch.cat(keys); torch.cuda.synchronize(); t1 = time.perf_counter()
    k, c = torch.unique(cat, return_counts=True); torch.cuda.synchronize(); t2 = time.perf_counter()
    kk, cc = k.cpu().numpy().tolist(), c.cpu().numpy().tolist(); t3 = time.perf_counter()
    cnt = Counter(); cnt.update(dict(zip(kk, cc))); t4 = time.perf_counter()
    print(f"cat {1e3*(t1-t0):.2f} ms, unique {1e3*(t2-t1):.2f} ms, to host {1e3*(t3-t2):.2f} ms, Counter {1e3*(t4-t3):.2f} ms")
