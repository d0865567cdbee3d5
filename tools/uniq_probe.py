import torch, time
torch.cuda.init()
for n in (131072, 524288, 2097152):
    k = torch.randint(0, 2**30, (n,), device="cuda", dtype=torch.int64) % 14400
    for name, fn in (("unique", lambda: torch.unique(k, return_counts=True)), ("sort", lambda: torch.sort(k)), ("bincount", lambda: torch.bincount(k, minlength=14400))):
        fn(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5): fn()
        torch.cuda.synchronize()
        print(n, name, f"{(time.perf_counter() - t0) / 5 * 1e3:.3f} ms")
