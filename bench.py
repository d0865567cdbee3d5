#!/usr/bin/env python3
"""bench.py — headline benchmark: unique samples/s through log-psi eval + E_loc on N2 (20 qubits),
M = 10 000 unique samples per GPU (BASELINE.json configs[1]).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one pass of the hot path over one batch of M unique sampled bit-strings that is
already resident in HBM: (log-psi evaluation of the batch ->) hash build -> matrix-free E_loc ->
weighted energy accumulators.  The K timed steps are K independent batches; two are in flight at
a time (`--pipeline`, two HIP streams) so that the E_loc / reduce kernels of one batch overlap the
log-psi kernel of the next; the serial figures are measured in the same run (`serial`,
`roofline.isolated`).  With N > 1 every rank owns an independent batch stream (weak scaling) and the
per-step energy accumulators [K, 4] are summed over the ranks by one RCCL all-reduce at the end of
the timed region — the only collective on the path.
Prints ONE JSON line on rank 0 (contract in the task statement) incl. `roofline` and
`cpu_baseline`.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, "naqs-for-quantum-chemistry_amd")
for _p in (ROOT, PKG):
    if _p not in sys.path:
        sys.path.insert(0, _p)

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s achievable
PROF_STRIDE = 10
MFMA_BF16_PEAK_TF = 2500.0  # dense bf16 MFMA peak (MI355X_MICROARCH.md)
MFMA_F32_PEAK_TF = 157.3   # f32-in/f32-acc MFMA dense peak (MI355X_MICROARCH.md: = the f32 vector rate)


def physical_keys(n_qubits, n_alpha, n_beta):
    from itertools import combinations
    al = [sum(1 << b for b in c) for c in combinations(range(0, n_qubits, 2), n_alpha)]
    be = [sum(1 << b for b in c) for c in combinations(range(1, n_qubits, 2), n_beta)]
    return np.sort(np.array([a | b for a in al for b in be], np.uint64))


def make_batch(ham, M, seed):
    """keys = sort(RandomState(1234).choice(all physical keys, M)); synthetic
    psi: log|psi| ~ N(-ln(M)/2, 2), phase ~ U[0, 2pi)."""
    n_orb = ham.n_qubits // 2
    from math import comb
    if comb(n_orb, ham.n_alpha) * comb(n_orb, ham.n_beta) <= 2_000_000:
        space = physical_keys(ham.n_qubits, ham.n_alpha, ham.n_beta)
        keys = np.sort(np.random.RandomState(1234 + seed).choice(space, M, replace=False))
    else:
        # config C4 (Li2O: 41 409 225 states): M distinct keys, alpha part = random n_alpha-subset of the even
        # bits, beta part = random n_beta-subset of the odd bits
        rs0 = np.random.RandomState(1234 + seed)
        ev, od, out = np.arange(0, ham.n_qubits, 2), np.arange(1, ham.n_qubits, 2), set()
        while len(out) < M:
            a, b_ = rs0.choice(ev, ham.n_alpha, replace=False), rs0.choice(od, ham.n_beta, replace=False)
            out.add(int(sum(1 << int(q) for q in a) | sum(1 << int(q) for q in b_)))
        keys = np.sort(np.array(list(out), np.uint64))
    rs = np.random.RandomState(4321 + seed)
    log_psi = np.stack([rs.normal(-0.5 * np.log(M), 2.0, M), rs.uniform(0, 2 * np.pi, M)], -1).astype(np.float32)
    counts = rs.poisson(5, M) + 1
    return keys, log_psi, counts


def algorithmic_bytes(M, K, Kxy):
    """per sample 8 B key + 16 B psi in, 16 B E_loc out, and per candidate connection
    one 8 B key probe + one 16 B psi fetch; the packed term table once per launch."""
    return M * (40 + 24 * Kxy) + 16 * K + 12 * Kxy


def cpu_baseline(ham_p, keys, log_psi, wf_args, budget_s=12.0):
    """CPU leg, same two stages as the GPU step, on the host cores of this box, bounded sample:
      * E_loc: the oracle's staged restatement of the reference algorithm (update_H + get_H + SpMV
        with a cold Hamiltonian cache — equal work to the matrix-free GPU path), OpenMP;
      * log-psi eval: the same torch modules the reference would run on CPU (its nade.py is PyTorch),
        float32, torch's default intra-op threads."""
    import torch
    from naqs_amd.hilbert import Encoding, Hilbert
    from naqs_amd.wavefunction import NAQSComplex_NADE_orbitals
    from oracle import oracle
    psi = np.exp(log_psi[:, 0].astype(np.float64)) * np.exp(1j * log_psi[:, 1].astype(np.float64))
    Ms = min(len(keys), 4000)          # M*Kyz / M*Kxy temporaries like the reference; bounded
    k, p = keys[:Ms], psi[:Ms]
    threads = oracle.max_threads()
    args = (ham_p.n_qubits, ham_p.n_alpha, ham_p.n_beta, ham_p.xy, ham_p.yz, ham_p.coeff, k, p)
    oracle.eloc_staged(*args)
    reps, t0 = 0, time.perf_counter()
    while True:
        oracle.eloc_staged(*args)
        reps += 1
        dt = time.perf_counter() - t0
        if dt > budget_s / 2 or reps >= 100:
            break
    t_eloc = dt / reps
    hil = Hilbert.get(ham_p.n_qubits, ham_p.n_alpha, ham_p.n_beta, encoding=Encoding.SIGNED)
    wf = NAQSComplex_NADE_orbitals(hil, device="cpu", **wf_args)
    states = hil.idx2state(torch.from_numpy(k.astype(np.int64)))
    with torch.no_grad():
        wf.log_psi(states)
        reps2, t0 = 0, time.perf_counter()
        while True:
            wf.log_psi(states)
            reps2 += 1
            dt2 = time.perf_counter() - t0
            if dt2 > budget_s / 2 or reps2 >= 100:
                break
    t_lp = dt2 / reps2
    return {"value": Ms / (t_eloc + t_lp), "unit": "unique samples/s", "cores": int(threads), "kind": "port",
            "sample": f"first {Ms} samples of the N2 batch: {reps} x E_loc (oracle staged restatement of "
                      f"update_H+get_H+SpMV, cold cache, {threads} OpenMP threads, {t_eloc * 1e3:.1f} ms each) + "
                      f"{reps2} x log-psi eval (torch CPU float32, {torch.get_num_threads()} threads, "
                      f"{t_lp * 1e3:.1f} ms each)",
            "eloc_only_samples_per_s": Ms / t_eloc, "logpsi_only_samples_per_s": Ms / t_lp}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=400)
    ap.add_argument("--warmup", type=int, default=40)
    ap.add_argument("--molecule", default="N2")
    ap.add_argument("--samples", type=int, default=10000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--serial-segment", action="store_true",
                    help="after the timed region, run 200 more steps one batch at a time and report them under `serial`")
    ap.add_argument("--pipeline", type=int, default=2,
                    help="independent batches in flight (HIP streams, one Hamiltonian/network handle pair each); 1 = serial")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from naqs_amd import hamiltonian, packing

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1 or os.environ.get("NAQS_BENCH_FORCE_DIST") == "1":
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=rank, world_size=world)

    ham_p = packing.load_packed(os.path.join(ROOT, "tests", "golden", f"ham_{args.molecule}.npz"))
    ham = hamiltonian.DevicePauliHamiltonian(ham_p, device=dev)
    M = args.samples
    keys_np, log_psi_np, counts_np = make_batch(ham_p, M, seed=rank)   # log_psi_np: CPU-baseline psi only
    keys = hamiltonian.keys_to_device(keys_np, dev)
    # ansatz of the published runs (experiments/bash/naqs/batch_train.sh:14): amplitude blocks 1x64,
    # one phase block 2x512, random init (no checkpoints without network) -> log psi of the batch
    from naqs_amd.hilbert import Encoding, Hilbert
    from naqs_amd.wavefunction import NAQSComplex_NADE_orbitals
    torch.manual_seed(1234 + rank)
    hil = Hilbert.get(ham_p.n_qubits, ham_p.n_alpha, ham_p.n_beta, encoding=Encoding.SIGNED)
    wf_args = dict(qubit_ordering=-1, amp_hidden_size=[64], phase_hidden_size=[512, 512], use_amp_spin_sym=True,
                   use_phase_spin_sym=False, aggregate_phase=False, n_alpha_electrons=ham_p.n_alpha,
                   n_beta_electrons=ham_p.n_beta)
    wf = NAQSComplex_NADE_orbitals(hil, device=dev, **wf_args)
    from naqs_amd.fused import FusedLogPsi
    # Throughput of a STREAM of independent batches: `depth` of them are in flight, each on its own HIP stream with its
    # own handle pair (a handle owns per-call scratch: hash table, psi table), so the E_loc / reduce kernels of one batch
    # run beside the log-psi kernel of the next (which leaves 47 of 256 CUs idle on its own).  --pipeline 1 = one batch
    # at a time; its step time is measured too and reported as `serial_ms_per_step`.
    depth = max(1, args.pipeline)
    hams = [ham] + [hamiltonian.DevicePauliHamiltonian(ham_p, device=dev) for _ in range(depth - 1)]
    nets = [FusedLogPsi(wf) for _ in range(depth)]       # libnaqs_hip.so: MFMA log-psi kernel
    fused = nets[0]
    streams = [torch.cuda.Stream(device=dev) for _ in range(depth)] if depth > 1 else [torch.cuda.current_stream(dev)]
    log_psis = [torch.empty((M, 2), dtype=torch.float32, device=dev) for _ in range(depth)]
    elocs = [torch.empty((M, 2), dtype=torch.float64, device=dev) for _ in range(depth)]
    log_psi = log_psis[0]
    weights = torch.as_tensor(counts_np / counts_np.sum(), dtype=torch.float64, device=dev)
    for h_ in hams:
        h_.reserve(M)
    torch.cuda.synchronize()
    # energy accumulators: two buffers, so that the RCCL all-reduce of step k (4 doubles, latency-bound on xGMI) runs on
    # the collective stream under the kernels of step k+1 instead of in front of them
    # energy accumulators: one row of 4 doubles per step, written by the reduce kernel of that step; with N > 1 GPUs the
    # rows of the whole timed region are summed over the ranks by ONE RCCL all-reduce at its end (few, larger
    # collectives: a per-step 32-byte all-reduce is pure xGMI latency and only a training step needs <E> that early)
    acc_all = torch.zeros((args.warmup + args.steps + 201, 4), dtype=torch.float64, device=dev)
    use_dist = world > 1 or os.environ.get("NAQS_BENCH_FORCE_DIST") == "1"      # (forced at world 1: exercises the path)
    n_done = [0]

    def step(d_override=None):
        row = acc_all[n_done[0]]
        d = (n_done[0] % depth) if d_override is None else d_override
        with torch.cuda.stream(streams[d]):
            # one library call: phase kernel (amplitude conditionals + phase MLP on the matrix cores; builds the key hash
            # and psi in f64) -> eloc_kernel -> reduce_kernel
            nets[d].log_psi_and_local_energy(hams[d], keys, weights=weights, log_psi_out=log_psis[d], eloc_out=elocs[d],
                                             sums_out=row)
        n_done[0] += 1

    def fence(first_row=None):
        torch.cuda.synchronize()                 # every stream of this rank
        if use_dist:
            if first_row is not None:
                dist.all_reduce(acc_all[first_row:n_done[0]])
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence(first_row=0)                           # (also sets the communicator up outside the timed region)
    # kernel durations: hipEvent pairs on the launch stream around every PROF_STRIDE-th launch of the timed
    # region (an event pair costs ~4 us of queue time; recording all of them slows the step by ~15 %)
    stride = max(1, min(PROF_STRIDE, args.steps // 8))
    for h_, n_ in zip(hams, nets):
        h_.prof_enable(args.steps // stride + 1, stride)
        n_.prof_enable(args.steps // stride + 1, stride)
    first_timed = n_done[0]
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence(first_row=first_timed)
    dt = time.perf_counter() - t0
    last_row = n_done[0] - 1
    kern_ms = launches = mlp_ms = mlp_launches = 0
    for h_, n_ in zip(hams, nets):
        a, b = h_.prof_read(); kern_ms += a; launches += b
        h_.prof_enable(0)
        a, b = n_.prof_read(); mlp_ms += a; mlp_launches += b
        n_.prof_enable(0)
    # the same K steps one batch at a time (single stream, first handle pair): latency of a batch, and the kernels'
    # durations when each has the GPU to itself
    serial = None
    if depth > 1 and world == 1 and args.serial_segment:
        ks = min(args.steps, 200)
        hams[0].prof_enable(ks // stride + 1, stride)
        nets[0].prof_enable(ks // stride + 1, stride)
        t1 = time.perf_counter()
        for _ in range(ks):
            step(0)
        fence()
        dts = time.perf_counter() - t1
        e_ms, e_n = hams[0].prof_read(); hams[0].prof_enable(0)
        p_ms, p_n = nets[0].prof_read(); nets[0].prof_enable(0)
        serial = {"ms_per_step": dts / ks * 1e3, "steps": ks, "eloc_kernel_us": e_ms / max(e_n, 1) * 1e3,
                  "logpsi_kernel_us": p_ms / max(p_n, 1) * 1e3}

    t = torch.tensor([dt], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t.item())

    if rank == 0:
        s = acc_all[last_row].cpu().numpy()
        b_alg = algorithmic_bytes(M, ham.K, ham.Kxy)
        t_kernel = kern_ms / max(launches, 1) * 1e-3
        achieved = b_alg / t_kernel / 1e9 if t_kernel > 0 else 0.0
        eloc_roof = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": None, "kernel": "eloc_kernel",
                     "kernel_us": t_kernel * 1e6, "algorithmic_bytes_per_launch": b_alg}
        # phase MLP: 2*(K*N) flops per layer and sample (18->512->512->4 for N2), f32 matrix cores
        dims = [max(1, 2 * (ham.n_qubits // 2 - 1)), 512, 512, 4]
        flops = 2.0 * M * sum(a * b for a, b in zip(dims, dims[1:]))
        # ... and, unless NAQS_AMP_MODE=0 keeps them in their own kernel, the amplitude blocks (pair n: 2n -> 64 -> 5)
        # evaluated in the same launch (2 * sum_n (max(1, 2n) * 64 + 64 * 5) flops per sample)
        amp_in_kernel = os.environ.get("NAQS_AMP_MODE", "1") == "1" and os.environ.get("NAQS_PHASE_MODE", "1") == "1"
        if amp_in_kernel:
            flops += 2.0 * M * sum(max(1, 2 * n) * 64 + 64 * 5 for n in range(ham.n_qubits // 2))
        t_mlp = mlp_ms / max(mlp_launches, 1) * 1e-3
        mlp_tf = flops / t_mlp / 1e12 if t_mlp > 0 else 0.0
        # The kernel evaluates the f32 network with every operand split into three bf16 planes (six exact cross
        # products per multiply on the bf16 matrix cores, f32 accumulate): f32-equivalent results (tests/
        # test_nade_gpu.py compares against float64).  `achieved` = ALGORITHMIC f32 flops / time, priced against
        # the dense f32-MFMA peak (the precision class of the computation); the executed-instruction view (6x
        # as many bf16 flops against the 2.5 PFLOP/s bf16 peak) is given next to it.
        bf16_mode = os.environ.get("NAQS_PHASE_MODE", "1") == "1"
        mlp_roof = {"bound": "mfma", "achieved": mlp_tf, "peak": MFMA_F32_PEAK_TF, "unit": "TFLOP/s",
                    "frac": mlp_tf / MFMA_F32_PEAK_TF, "traffic": None,
                    "kernel": ("phase_kernel_bf16x3 (" + ("amplitude conditionals + " if amp_in_kernel else "") +
                               "phase MLP; f32 via 3-way bf16 split, v_mfma_f32_16x16x32_bf16)") if bf16_mode
                              else "phase_kernel (f32 MFMA 16x16x4)",
                    "kernel_us": t_mlp * 1e6, "algorithmic_flops_per_launch": flops}
        if bf16_mode:
            mlp_roof["executed"] = {"dtype": "bf16", "tflops": 6 * mlp_tf, "peak": MFMA_BF16_PEAK_TF,
                                    "frac": 6 * mlp_tf / MFMA_BF16_PEAK_TF}
        # HBM bytes per launch from the committed rocprofv3 PMC passes of this same command
        # (tools/collect_pmc.py; FETCH_SIZE/WRITE_SIZE in separate passes, gfx950 corrections applied there)
        try:
            with open(os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")) as f:
                pmc = json.load(f)
            if args.molecule == "N2" and M == 10000:           # the PMC passes were taken on this workload
                eloc_roof["traffic"] = pmc["eloc_kernel"]["hbm_bytes_per_launch"]
                mlp_roof["traffic"] = pmc["phase_kernel"]["hbm_bytes_per_launch"]
        except (OSError, KeyError, ValueError):
            pass
        if serial is None and depth > 1 and args.molecule == "N2" and M == 10000:
            # kernel durations of `bench.py --pipeline 1` on this workload, from the committed run (profiles/)
            try:
                with open(os.path.join(ROOT, "profiles", "r01_bench_n2_10k_serial.json")) as f:
                    ser = json.load(f)
                serial = {"ms_per_step": ser["ms_per_step"], "steps": ser["steps"], "source": "profiles/r01_bench_n2_10k_serial.json",
                          "logpsi_kernel_us": ser["roofline"]["kernel_us"],
                          "eloc_kernel_us": ser["roofline"]["other_kernels"][0]["kernel_us"]}
            except (OSError, KeyError, ValueError, IndexError):
                serial = None
        if serial is not None:
            # the same kernels with the GPU to themselves (one batch at a time): what the kernel itself achieves; the
            # durations above are longer because the next batch's kernels share the CUs during the timed region
            if serial["logpsi_kernel_us"] > 0:
                tf = flops / (serial["logpsi_kernel_us"] * 1e-6) / 1e12
                mlp_roof["isolated"] = {"kernel_us": serial["logpsi_kernel_us"], "achieved": tf, "frac": tf / MFMA_F32_PEAK_TF}
            if serial["eloc_kernel_us"] > 0:
                gb = b_alg / (serial["eloc_kernel_us"] * 1e-6) / 1e9
                eloc_roof["isolated"] = {"kernel_us": serial["eloc_kernel_us"], "achieved": gb, "frac": gb / HBM_PEAK_GBS}
        dominant, other = (mlp_roof, eloc_roof) if t_mlp >= t_kernel else (eloc_roof, mlp_roof)
        roofline = dict(dominant)
        roofline["other_kernels"] = [other]
        out = {
            "metric": f"unique samples/sec through E_loc + log-psi eval ({args.molecule}, {ham.n_qubits} qubits)",
            "value": world * M * args.steps / dt,
            "unit": "unique samples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32 network (bf16x3-split MFMA, f32-equivalent) / f64 E_loc", "data": "synthetic",
            "config": {"workload": f"{args.molecule} STO-3G ({ham.n_qubits} qubits, K={ham.K} Pauli terms, "
                                   f"Kxy={ham.Kxy}), {M} unique samples per GPU, 1xMI355X per rank",
                       "stages": "fused NADE log-psi eval (amp 1x64, phase 2x512; builds the key hash + psi table) + matrix-free E_loc "
                                 "(f64) + weighted energy reduction"
                                 + (" + one RCCL all-reduce of the per-step accumulators [K, 4] at the end of the timed region"
                                    if world > 1 else ""),
                       "pipeline": (f"{depth} independent batches in flight on {depth} HIP streams (one handle pair each)"
                                    if depth > 1 else "one batch at a time"),
                       "input": "unique sampled bit-strings (keys + int8 occupations) resident in HBM; random-init network",
                       "energy": float(s[0] / s[3])},
            "roofline": roofline,
        }
        if serial is not None:
            out["serial"] = serial
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(ham_p, keys_np, log_psi_np, wf_args)
    else:
        out = None
    if dist.is_initialized():
        dist.destroy_process_group()
    if out is not None:
        # the JSON line must be the last line of stdout: RCCL writes its version banner through C stdio, which would
        # otherwise be flushed at exit, after Python's own output
        import ctypes
        try:
            ctypes.CDLL(None).fflush(None)
        except OSError:
            pass
        sys.stdout.flush()
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
