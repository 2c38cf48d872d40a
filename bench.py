#!/usr/bin/env python3
"""bench.py -- SNP-updates/sec of one E-step sweep over synthetic LD blocks on N MI355X GPUs.

Contract (driver): `python bench.py --gpus N --steps K --warmup W`; for N > 1 it is launched as
`python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...`, one rank per GPU.
Rank 0 prints ONE JSON line.

A "step" = one pass of the hot path over one batch: the variational state is re-initialised on
the device to the standard start (var_gamma = pi, var_mu = eta = q = eta_diff = 0; the reference's
own benchmark re-initialises before every timed call, benchmarks/benchmark_e_step.py:58-61,
because a converged state takes the skip branch e_step.hpp:410-413 and reads no LD) and one
E-step sweep runs over every LD block of the workload.  LD and the per-SNP inputs are resident in
HBM before the timed region starts.

Workload (N = 1): BASELINE.json configs[2] -- ~1.1 M SNPs in ~1 700 LD blocks (lognormal block
sizes, AR(1) LD), spike-and-slab prior, fp32 state, fp32 LD, symmetric form.
N > 1: "weak" (default) gives every rank its own genome-scale workload (blocks are independent,
no data-path collective); "strong" shards the blocks of ONE workload over the ranks.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X HBM3E peak (MI355X_MICROARCH.md: 8 TB/s spec, ~6.3 achievable)
STATE_BYTES_PER_SNP = 68     # lb+indptr (12) + 4 inputs (16) + 5 state reads/writes (40)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", default="cfg3", choices=["cfg1", "cfg2", "cfg3"])
    ap.add_argument("--low-memory", action="store_true", help="upper-triangular LD (reference default)")
    ap.add_argument("--ld-dtype", default="float32", choices=["float32", "int8", "int16"])
    ap.add_argument("--math", default="exact", choices=["exact", "fast"])
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"])
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="budget of the CPU baseline leg (0 = skip)")
    ap.add_argument("--model", default="spike_slab", choices=["spike_slab", "mixture", "grid"],
                    help="spike_slab = the headline (configs[1..2]); mixture = configs[3] (VIPRSMix); grid = configs[4]")
    ap.add_argument("--width", type=int, default=0, help="mixture components K (default 4) / grid models G (default 32)")
    ap.add_argument("--seed", type=int, default=7209)
    return ap.parse_args()


def shard_blocks_lpt(sizes, n_parts):
    """Longest-processing-time bin packing of blocks (cost ~ b^2) over ranks (strong scaling)."""
    order = np.argsort(-sizes, kind="stable")
    load = np.zeros(n_parts)
    parts = [[] for _ in range(n_parts)]
    for i in order:
        r = int(np.argmin(load))
        parts[r].append(int(i))
        load[r] += float(sizes[i]) ** 2
    return [sorted(p) for p in parts]


def cpu_baseline(ld, inp, budget_s, model="spike_slab", width=1, extra=None, pi0=None):
    """The reference's own e_step.hpp (oracle/_ref, built from /root/reference by oracle/Makefile)
    timed on this host, state re-initialised before every call: first with threads=1 (the parity
    reference), then with its OpenMP path on all cores (racy Hogwild, e_step.hpp:384-387 -- the
    "reference multithreaded-CPU" figure).  Sample: leading blocks of the same workload sized to
    the time budget (a mixture / grid SNP-update costs `width` times a spike-and-slab one)."""
    from oracle import oracle as O
    kind = "reference" if O.have_reference() else "restated"
    cores = os.cpu_count() or 1
    # ~0.2-0.5 M SNP-updates/s single-threaded: size the sample for ~budget/3 per single-thread pass
    target_snps = int(min(ld.m, max(2000, 0.1e6 * budget_s / max(1, width))))
    nb = int(np.searchsorted(ld.block_start, target_snps, side="left"))
    nb = max(1, min(nb, len(ld.block_start) - 1))
    m_s = int(ld.block_start[nb])
    nnz_s = int(ld.ld_indptr[m_s])
    lb = np.ascontiguousarray(ld.ld_left_bound[:m_s])
    ip = np.ascontiguousarray(ld.ld_indptr[:m_s + 1])
    data = ld.ld_data[:nnz_s]
    T = np.float32
    std_beta = np.ascontiguousarray(inp.std_beta[:m_s])
    if model == "spike_slab":
        vec = {k: np.ascontiguousarray(getattr(inp, k)[:m_s]) for k in ("u_logs", "sqrt_half_var_tau", "mu_mult")}
    else:
        order = "F" if model == "grid" else "C"
        vec = {k: np.asarray(v[:m_s], order=order).copy(order=order) for k, v in extra.items()}

    def one(threads):
        if model == "spike_slab":
            st = {k: np.ascontiguousarray(v[:m_s]).copy() for k, v in inp.state_copy().items()}
            t0 = time.perf_counter()
            O.cpp_e_step(lb, ip, data, std_beta, st["var_gamma"], st["var_mu"], st["eta"], st["q"],
                         st["eta_diff"], vec["u_logs"], vec["sqrt_half_var_tau"], vec["mu_mult"], ld.dq_scale,
                         threads, ld.low_memory, kind=kind)
        elif model == "mixture":
            vg = np.full((m_s, width), pi0, dtype=T)
            vm = np.zeros((m_s, width), dtype=T)
            eta, q, ed = (np.zeros(m_s, dtype=T) for _ in range(3))
            t0 = time.perf_counter()
            O.cpp_e_step_mixture(lb, ip, data, std_beta, vg, vm, eta, q, ed, vec["log_null_pi"], vec["u_logs"],
                                 vec["sqrt_half_var_tau"], vec["mu_mult"], ld.dq_scale, threads, ld.low_memory, kind=kind)
        else:
            vg = np.full((m_s, width), pi0, dtype=T, order="F")
            vm, eta, q, ed = (np.zeros((m_s, width), dtype=T, order="F") for _ in range(4))
            t0 = time.perf_counter()
            O.cpp_e_step_grid(lb, ip, data, std_beta, vg, vm, eta, q, ed, vec["u_logs"], vec["half_var_tau"],
                              vec["mu_mult"], ld.dq_scale, np.arange(width, dtype=np.int32), threads, ld.low_memory,
                              kind=kind)
        return time.perf_counter() - t0

    res = {}
    for name, threads in (("threads1", 1), ("all_cores", cores if kind == "reference" else 1)):
        one(threads)                                   # warm-up
        ts, t_used = [], 0.0
        while t_used < budget_s / 3 and len(ts) < 15:
            dt = one(threads)
            ts.append(dt)
            t_used += dt
        res[name] = m_s / float(np.median(ts))
    return {
        "value": res["all_cores"], "unit": "SNP-updates/s", "cores": cores if kind == "reference" else 1,
        "kind": "reference" if kind == "reference" else "port",
        "sample": f"first {nb} LD blocks ({m_s} SNPs, {nnz_s} LD entries) of the same workload, model={model}"
                  + (f" width={width}" if model != "spike_slab" else "")
                  + f", state re-initialised per call, median; OpenMP threads={cores} (racy, as the reference)",
        "single_thread_value": res["threads1"],
    }


def main():
    args = parse_args()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    n_gpus = args.gpus
    if world != n_gpus and world > 1:
        raise SystemExit(f"--gpus {n_gpus} but WORLD_SIZE={world}")

    dist = None
    backend = os.environ.get("VIPRS_BENCH_BACKEND", "nccl")       # nccl = RCCL over xGMI; gloo for dry runs
    if world > 1:
        import torch
        import torch.distributed as dist
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=backend)

    from viprs_amd import _lib
    from viprs_amd.plan import DeviceState, LDPlan
    from viprs_amd.utils import synthetic as syn

    if _lib.device_count() < 1:
        raise SystemExit("bench.py needs a HIP device (the E-step has no CPU fallback)")

    # ---- workload -------------------------------------------------------------------------------
    ld_dtype = np.dtype(args.ld_dtype)
    sizes_all = syn.block_sizes(args.config, args.seed)
    if world > 1 and args.scaling == "strong":
        mine = shard_blocks_lpt(sizes_all, world)[rank]
        sizes = sizes_all[mine]
        seed = args.seed
    else:
        sizes = sizes_all
        seed = args.seed + 1000 * rank          # weak: every rank its own genome-scale workload
    ld = syn.make_ld(sizes, low_memory=args.low_memory, ld_dtype=ld_dtype, seed=seed)
    ss = syn.make_sumstats(ld, seed=seed)
    inp = syn.make_inputs(ss)

    device = local_rank % _lib.device_count()
    plan = LDPlan(ld.ld_left_bound, ld.ld_indptr, ld.ld_data, ld.low_memory, device=device, math_mode=args.math)
    width = args.width or {"spike_slab": 1, "mixture": 4, "grid": 32}[args.model]
    state = DeviceState(plan, "float32", args.model, width)
    active = None
    host_extra = None
    pi0 = inp.pi
    if args.model == "spike_slab":
        for name in ("std_beta", "u_logs", "sqrt_half_var_tau", "mu_mult"):
            state.upload(name, getattr(inp, name))
    else:
        extra = syn.make_mixture_inputs(ss, width) if args.model == "mixture" else syn.make_grid_inputs(ss, width)
        pi0 = extra.pop("pi")
        host_extra = extra
        state.upload("std_beta", inp.std_beta)
        for name, arr in extra.items():
            state.upload(name, arr)
        if args.model == "grid":
            active = np.arange(width, dtype=np.int32)

    def step():
        state.reset(pi0)
        state.e_step(ld.dq_scale, active, sync=False)

    def barrier():
        state.synchronize()
        if dist is not None:
            if backend == "nccl":
                import torch
                torch.cuda.synchronize()
            dist.barrier()

    for _ in range(args.warmup):
        step()
    barrier()
    plan.timing_reset()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    state.synchronize()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        import torch
        dev = "cuda" if backend == "nccl" else "cpu"
        if backend == "nccl":
            torch.cuda.synchronize()
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        tot = torch.tensor([float(ld.m)], dtype=torch.float64, device=dev)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        total_snps = float(tot.item())
    else:
        total_snps = float(ld.m)

    skipped = plan.last_skipped()
    k_ms = plan.timing_history(which=1)
    sweep_ms = plan.timing_history(which=0)

    if rank == 0:
        es = ld_dtype.itemsize
        nnz_streamed = int(ld.ld_indptr[-1]) * (2 if ld.low_memory else 1)   # upper form is read twice
        # state bytes per SNP: index (12) + per model column 4 inputs + 5 state reads/writes + std_beta
        state_bytes = STATE_BYTES_PER_SNP if args.model == "spike_slab" else (
            12 + 4 + 4 * (3 * width + 1) + 8 * (2 * width + 3) if args.model == "mixture" else 12 + 4 + 36 * width)
        algo_bytes = es * nnz_streamed + state_bytes * ld.m
        k_avg_ms = float(np.mean(k_ms)) if k_ms else float("nan")
        achieved = algo_bytes / (k_avg_ms * 1e-3) / 1e9
        traffic = None
        prof = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(prof):
            try:
                key = f"{args.config}_{args.ld_dtype}_{'upper' if args.low_memory else 'sym'}"
                if args.model != "spike_slab":
                    key += f"_{args.model}{width}"
                traffic = json.load(open(prof)).get(key)
            except Exception:
                traffic = None
        out = {
            "metric": "SNP-updates/sec/E-step (1M SNPs, ~1700 LD blocks)",
            "value": total_snps * args.steps / elapsed,
            "unit": "SNP-updates/s",
            "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": args.scaling if n_gpus > 1 else "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": {"cfg1": "configs[0]: single LD block, 500 SNPs",
                             "cfg2": "configs[1]: chr22-like, ~19k SNPs / 40 LD blocks",
                             "cfg3": "configs[2]: genome-wide, ~1.1M SNPs / 1700 LD blocks"}[args.config]
                            + {"spike_slab": ", spike-and-slab", "mixture": f", sparse mixture prior K={width} (configs[3])",
                               "grid": f", grid of {width} (pi, sigma_eps) models batched per SNP (configs[4])"}[args.model]
                            + ", AR(1) block LD",
                "prior": args.model, "prior_width": width,
                "snp_x_grid_point_updates_per_s": total_snps * width * args.steps / elapsed if args.model == "grid" else None,
                "snps_per_gpu": int(ld.m), "ld_blocks_per_gpu": int(len(sizes)),
                "ld_entries_per_gpu": int(ld.ld_indptr[-1]), "ld_dtype": args.ld_dtype,
                "ld_form": "upper-triangular (low_memory=True)" if ld.low_memory else "symmetric (low_memory=False)",
                "math_mode": args.math, "skipped_snps_last_sweep": int(skipped),
                "parallelism": f"ld-blocks x{n_gpus} ({args.scaling})" if n_gpus > 1 else "single GPU",
                "step": "device state re-init + one E-step sweep over all blocks",
            },
            "roofline": {
                "bound": "hbm",
                "kernel": ("estep_grid_mfma_kernel (one workgroup per LD block, LD read once for all models)"
                           + (" + estep_grid_upper_epilogue_kernel" if ld.low_memory else "")) if args.model == "grid" else
                          ("estep_panel_kernel (3 size classes on 3 streams; the largest two share blocks between CUs)"
                           + (" + estep_upper_epilogue_kernel" if ld.low_memory else "")),
                "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                "traffic": traffic, "algorithmic_bytes_per_launch": int(algo_bytes),
                "kernel_ms_avg": k_avg_ms, "sweep_ms_avg": float(np.mean(sweep_ms)) if sweep_ms else None,
            },
        }
        if n_gpus == 1 and args.cpu_seconds > 0:
            out["cpu_baseline"] = cpu_baseline(ld, inp, args.cpu_seconds, args.model, width, host_extra, pi0)
        print(json.dumps(out), flush=True)

    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
